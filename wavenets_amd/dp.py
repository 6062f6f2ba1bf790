"""Data-parallel step helpers (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on MI355X, "gloo" in the CPU tests).

The reference trains under ``tf.distribute.MirroredStrategy`` (train.py:203): every replica
computes gradients of ``sum_local(loss) / B_global`` (tf.nn.compute_average_loss,
src/model.py:328-329) and the optimizer SUM-all-reduces them (src/model.py:336).  Here the
same exchange is ONE all-reduce over the flat fp32 gradient bucket (gradient + {loss, reg_loss}; 5.0 MB for
BASELINE configs[2]); utterances are sharded by rows, parameters and Adam state replicated.
Clipnorm is applied AFTER the reduction on every rank (deterministic, identical replicas;
SURVEY.md section 8c records the Keras-version ambiguity)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def initialized() -> bool:
  """A process group exists (world size 1 included): the data-parallel step then goes through the collective backend --
  RCCL for backend "nccl" -- even when this process is the only replica."""
  return dist.is_available() and dist.is_initialized()


def world_size() -> int:
  return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
  return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_rows(global_batch: int, world: int, rnk: int) -> slice:
  """Rows of the global batch owned by rank ``rnk`` (contiguous, equal shares)."""
  if global_batch % world != 0:
    raise ValueError('global batch must be divisible by the number of replicas')
  per = global_batch // world
  return slice(rnk * per, (rnk + 1) * per)


def allreduce_gradients(flat_grads: torch.Tensor, loss: torch.Tensor, group=None) -> None:
  """In-place SUM all-reduce of the flat gradient bucket and of the {loss, reg_loss} pair.

  loss[0] holds this replica's ``sum_local(l) / B_global`` -> summed gives the global loss;
  loss[1] holds ``l2 / n_replicas`` on every replica -> summed gives the full penalty."""
  if not initialized():
    return
  dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
  dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=group)


def allreduce_bucket(bucket: torch.Tensor, group=None) -> None:
  """The same exchange as ``allreduce_gradients`` when the {loss, reg_loss} pair lives right behind the flat
  gradient in one buffer (``WaveNet._grad_bucket``): a single collective per step.  Without a process group: nothing;
  with one -- of any size, 1 included -- the backend's all-reduce runs."""
  if not initialized():
    return
  dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
