"""Multi-GPU plumbing of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

The path shards by query and needs exactly one collective: an all-gather of the per-rank bounding boxes
(6 floats = 24 B per rank) so that every rank quantises curve keys on the same grid.  Every rank then
builds the same index and answers a contiguous, 64-aligned shard of the curve-sorted queries
(pcpx_shard_range); rows land in disjoint slices of the output, so no gather of results is needed.
"""
import torch

from .index import shard_range


def input_slice(n, rank, world):
    """Points [lo, hi) of the input whose bounding box this rank computes."""
    return n * rank // world, n * (rank + 1) // world


def union_of_boxes(boxes):
    """boxes: (world, 6) tensor of {min xyz, max xyz}; returns the 6-float union."""
    return torch.cat([boxes[:, :3].min(0).values, boxes[:, 3:].max(0).values])


def global_grid(local_box, dist=None, world=1, always=False):
    """All-gather the per-rank boxes and return the union (identical on every rank).  always=True runs the collective
    even with one rank (a hardware rehearsal of RCCL initialisation and a device-tensor all-gather)."""
    if dist is None or (world <= 1 and not always):
        return local_box.clone()
    gathered = [torch.empty_like(local_box) for _ in range(world)]
    dist.all_gather(gathered, local_box)
    return union_of_boxes(torch.stack(gathered))


def query_shard(n_indexed, rank, world):
    """(first, count) in curve-sorted positions for this rank; first is a multiple of 64."""
    return shard_range(n_indexed, rank, world)
