"""CPU oracle (test infrastructure only; see wavenet_oracle.py header)."""
