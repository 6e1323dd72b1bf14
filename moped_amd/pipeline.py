"""Frame driver over the C ABI: device-resident MATCH -> CLUSTER -> POSE -> FILTER ->
POSE2 -> FILTER2, one mh_ctx per frame in flight, optional model sharding over
ranks with the two small exchanges of SURVEY.md 8(e).

PyTorch is used for what the C ABI does not own: device buffers for the frame
inputs, HIP streams, and torch.distributed (RCCL) for the all-gathers.  All
computation happens inside libmoped_hip.so.
"""
from __future__ import annotations

import numpy as np
import torch

from . import capi


class ShardedDB:
    """This rank's slice of the model database, models assigned in contiguous
    blocks: rank r owns models [r*n/W, (r+1)*n/W)."""

    def __init__(self, desc, xyz, model_of, n_models, rank=0, world=1):
        model_of = np.asarray(model_of, np.int32)
        lo_m = (rank * n_models) // world
        hi_m = ((rank + 1) * n_models) // world
        rows = np.nonzero((model_of >= lo_m) & (model_of < hi_m))[0]
        # rows of a model are contiguous in the flattened DB (MATCH_ANN_CPU::Update order)
        self.row_lo = int(rows[0]) if len(rows) else 0
        self.row_hi = int(rows[-1]) + 1 if len(rows) else 0
        self.desc = np.ascontiguousarray(desc[self.row_lo:self.row_hi], np.float32)
        self.xyz = np.ascontiguousarray(xyz[self.row_lo:self.row_hi], np.float32)
        self.model_of = np.ascontiguousarray(model_of[self.row_lo:self.row_hi])
        self.n_models = n_models          # model ids stay global
        self.rank, self.world = rank, world


class FramePipeline:
    """`depth` frames in flight on one GPU; each has its own mh_ctx + HIP stream so
    the latency-bound CLUSTER/POSE/FILTER kernels of one frame overlap the MATCH
    kernel of the next."""

    def __init__(self, device: int, db: ShardedDB, depth: int = 1, max_queries: int = 4096,
                 params: capi.mh_frame_params | None = None, K=None, cam=None, group=None,
                 force_exchange: bool = False):
        from . import synth
        self.dev = torch.device(f"cuda:{device}")
        torch.cuda.set_device(self.dev)
        self.db = db
        self.params = params or capi.default_frame_params()
        self.K = synth.K_DEFAULT if K is None else K
        self.cam = synth.CAM_IDENTITY if cam is None else cam
        self.group = group
        self.world = db.world
        self.ctxs, self.streams = [], []
        normalized = None
        for i in range(depth):
            c = capi.Context(device)
            s = torch.cuda.Stream(device=self.dev)
            c.set_stream(s.cuda_stream)   # before any work: the context then never creates a stream of its own
            if i == 0:
                # model descriptors are L2-normalised once, like Update() (MATCH_ANN_CPU.hpp:94)
                normalized = c.normalize(db.desc) if db.desc.shape[0] else db.desc
                c.db_upload(normalized, db.model_of, db.xyz, db.n_models, index_base=db.row_lo)
            else:
                c.db_share(self.ctxs[0])   # one store per GPU: every frame in flight searches the same copy
            c.reserve(max_queries)
            self.ctxs.append(c)
            self.streams.append(s)
        self.depth = depth
        self.exchange = force_exchange or self.world > 1
        self._local = [None] * depth    # send block per slot: [3][Q] words of exchange 1 + the result block riding along
        self._gather = [None] * depth   # receive block: W of those
        self._ex_q = [0] * depth
        self._inputs = [None] * depth   # the slot's current input tensors: its stream may still be reading them
        self._cam = capi.make_cam(self.K, self.cam)

    # Exchange 2 rides on exchange 1: behind its [3][Q] top-2 words every rank sends the result block
    # of the PREVIOUS frame of the same slot (complete by then: same stream), so one all-gather per
    # frame carries both exchanges of SURVEY 8(e).  EX2_OBJECTS objects per rank and frame.
    EX2_OBJECTS = 62
    EX2_WORDS = (16 + EX2_OBJECTS * capi.OBJECT_DTYPE.itemsize) // 4

    def _unpack_block(self, blk: np.ndarray) -> np.ndarray:
        """One rank's piggy-backed result block {n, flags, pad, pad, objects...} -> object array.
        The block carries at most EX2_OBJECTS objects: a frame with more, or one whose capacity flags
        are set, is an error here and not a silently shortened list (gather_objects() has no limit)."""
        n, flags = int(blk[0]), int(blk[1])
        if n > self.EX2_OBJECTS or flags != 0:
            raise RuntimeError(f"exchange 2: a rank reported {n} objects (block holds {self.EX2_OBJECTS}), "
                               f"capacity flags {flags}; use gather_objects() for this frame")
        return blk[4:].view(np.uint8)[:n * capi.OBJECT_DTYPE.itemsize].view(capi.OBJECT_DTYPE).copy()

    # ---- single frame in slot i ------------------------------------------------------
    def enqueue(self, slot: int, q_desc: torch.Tensor, q_uv: torch.Tensor, seed: int = 1,
                after: "torch.cuda.Stream | None" = None):
        """q_desc [Q,128] float32 (normalised in place), q_uv [Q,2]; both on this GPU.
        Returns immediately; work is on the slot's stream.  `after`: a stream whose
        pending work produces the inputs (omit when they are already resident --
        waiting on the legacy default stream serialises every frame behind it)."""
        c, s = self.ctxs[slot], self.streams[slot]
        Q = q_desc.shape[0]
        self._inputs[slot] = (q_desc, q_uv)   # kept until the slot's next enqueue (stream order: the old ones are done by then)
        if after is not None:
            s.wait_stream(after)
        if not self.exchange:
            c.frame_enqueue(q_desc.data_ptr(), q_uv.data_ptr(), Q, self.K, self.cam, self.params, seed)
            return
        stride = 3 * Q + self.EX2_WORDS
        if self._local[slot] is None or self._ex_q[slot] != Q:
            s.synchronize()   # an earlier frame of this slot may still read the old blocks
            self._local[slot] = torch.zeros(stride, dtype=torch.int32, device=self.dev)
            self._gather[slot] = torch.zeros(self.world * stride, dtype=torch.int32, device=self.dev)
            self._ex_q[slot] = Q
        local, gathered = self._local[slot], self._gather[slot]
        with torch.cuda.stream(s):
            c.frame_result_copy_dev(local.data_ptr() + 12 * Q, self.EX2_OBJECTS)   # exchange 2 of the slot's last frame
            c.frame_enqueue_match_local(q_desc.data_ptr(), Q, local.data_ptr())
            # exchange 1 (+2): every shard's per-query (idx1, d1, d2) -> [W][3][Q] (+ result blocks), one all-gather
            _all_gather_into(gathered, local, self.group)
            c.frame_enqueue_rest_strided(q_uv.data_ptr(), Q, gathered.data_ptr(), self.world, stride, self.K,
                                         self.cam, self.params, seed, _cam_struct=self._cam)

    # ---- batches of frames (small shards) ----------------------------------------------------
    def enqueue_batch(self, slot: int, q_desc: torch.Tensor, q_uv: torch.Tensor, B: int, seeds):
        """B frames through ONE MATCH launch and ONE exchange (a shard of a few thousand rows does not
        fill the chip for the 3000 queries of one frame): q_desc [B*Q,128], q_uv [B*Q,2], the frames one
        after the other; their CLUSTER..FILTER2 run one after the other on the slot's stream and leave
        their objects in result slots 0..B-1.  Needs the exchange path (sharded DB or force_exchange)."""
        assert self.exchange and 1 <= B <= capi.MAX_BATCH
        c, s = self.ctxs[slot], self.streams[slot]
        self._inputs[slot] = (q_desc, q_uv)
        BQ = q_desc.shape[0]
        Q = BQ // B
        tail = B * self.EX2_WORDS
        stride = 3 * BQ + tail
        if self._local[slot] is None or self._ex_q[slot] != -BQ:
            s.synchronize()
            self._local[slot] = torch.zeros(stride, dtype=torch.int32, device=self.dev)
            self._gather[slot] = torch.zeros(self.world * stride, dtype=torch.int32, device=self.dev)
            self._ex_q[slot] = -BQ          # negative: batch layout
            self._ex_b = B
        local, gathered = self._local[slot], self._gather[slot]
        with torch.cuda.stream(s):
            c.frame_result_copy_slots_dev(local.data_ptr() + 12 * BQ, B, self.EX2_OBJECTS)   # exchange 2 of the slot's last batch
            c.frame_enqueue_match_local(q_desc.data_ptr(), BQ, local.data_ptr())
            _all_gather_into(gathered, local, self.group)
            for f in range(B):
                c.frame_enqueue_rest_batch(q_uv.data_ptr() + 8 * f * Q, Q, gathered.data_ptr() + 4 * f * Q, self.world,
                                           stride, BQ, f, self.K, self.cam, self.params, int(seeds[f]),
                                           _cam_struct=self._cam)

    def fetch_batch(self, slot: int, B: int):
        return [self.ctxs[slot].frame_fetch_slot(f) for f in range(B)]

    def previous_objects_batch(self, slot: int):
        """Objects of the B frames of the batch enqueued in `slot` BEFORE the current one, from all ranks."""
        self.streams[slot].synchronize()
        BQ, B = -self._ex_q[slot], self._ex_b
        stride = 3 * BQ + B * self.EX2_WORDS
        host = self._gather[slot].view(self.world, stride)[:, 3 * BQ:].contiguous().cpu().numpy()
        out = []
        for f in range(B):
            objs = []
            for r in range(self.world):
                objs.append(self._unpack_block(host[r, f * self.EX2_WORDS:(f + 1) * self.EX2_WORDS]))
            out.append(np.concatenate(objs) if objs else np.zeros(0, capi.OBJECT_DTYPE))
        return out

    def flush_objects_batch(self, slot: int, B: int):
        """Exchange 2 for the LAST batch of a slot (nothing follows to carry it): one small all-gather."""
        c, s = self.ctxs[slot], self.streams[slot]
        with torch.cuda.stream(s):
            mine = torch.zeros(B * self.EX2_WORDS, dtype=torch.int32, device=self.dev)
            c.frame_result_copy_slots_dev(mine.data_ptr(), B, self.EX2_OBJECTS)
            out = _all_gather_flat(mine, self.world, self.group).view(self.world, B * self.EX2_WORDS)
        s.synchronize()
        host = out.cpu().numpy()
        res = []
        for f in range(B):
            objs = []
            for r in range(self.world):
                objs.append(self._unpack_block(host[r, f * self.EX2_WORDS:(f + 1) * self.EX2_WORDS]))
            res.append(np.concatenate(objs) if objs else np.zeros(0, capi.OBJECT_DTYPE))
        return res

    def previous_objects(self, slot: int):
        """Objects of the frame enqueued in `slot` BEFORE the current one, from all ranks, as they
        arrived with the current frame's exchange (no collective of its own).  Synchronises the slot."""
        self.streams[slot].synchronize()
        Q = self._ex_q[slot]
        stride = 3 * Q + self.EX2_WORDS
        host = self._gather[slot].view(self.world, stride)[:, 3 * Q:].contiguous().cpu().numpy()
        objs = []
        for r in range(self.world):
            objs.append(self._unpack_block(host[r]))
        return np.concatenate(objs) if objs else np.zeros(0, capi.OBJECT_DTYPE)

    def fetch(self, slot: int):
        """Objects of the frame in `slot` (model ids global), counts[4]."""
        return self.ctxs[slot].frame_fetch()

    def gather_objects(self, slot: int):
        """Exchange 2: every rank's result block -> all ranks; returns the merged
        object array (rank order = model order)."""
        import torch.distributed as dist
        c, s = self.ctxs[slot], self.streams[slot]
        ptr, nbytes = c.frame_result_dev()
        with torch.cuda.stream(s):
            mine = _wrap_uint8(ptr, nbytes, self.dev)
            out = _all_gather_flat(mine, self.world, self.group).view(self.world, nbytes)
        s.synchronize()
        host = out.cpu().numpy()
        objs = []
        for r in range(self.world):
            n = int(host[r, :4].view(np.int32)[0])
            blk = host[r, 16:16 + n * capi.OBJECT_DTYPE.itemsize].view(capi.OBJECT_DTYPE)
            objs.append(blk.copy())
        return np.concatenate(objs) if objs else np.zeros(0, capi.OBJECT_DTYPE)

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        for c in self.ctxs:
            c.close()
        self.ctxs = []


def exchange_top2(local: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """Exchange 1: one fused all-gather of every rank's per-query
    local top-2.  local = [3][Q] int32 words (idx1, bits of d1, bits of d2);
    returns [W][3][Q], the block mh_frame_enqueue_rest takes.  Device-agnostic
    (RCCL on GPU, gloo in the CPU tests)."""
    Q = local.shape[-1] if local.dim() == 2 else local.numel() // 3
    out = torch.empty(world * 3 * Q, dtype=local.dtype, device=local.device)
    _all_gather_into(out, local.contiguous().view(-1), group)
    return out.view(world, 3, Q)


def _all_gather_into(out: torch.Tensor, mine: torch.Tensor, group=None) -> None:
    """all_gather_into_tensor on flat buffers.  RCCL takes device tensors directly;
    the gloo backend (CPU tests, and the 2-ranks-on-one-GPU rehearsal of the N > 1
    path) is fed through host memory."""
    import torch.distributed as dist
    if mine.is_cuda and dist.get_backend(group) == "gloo":
        host = mine.cpu()
        tmp = torch.empty(out.numel(), dtype=host.dtype)
        dist.all_gather_into_tensor(tmp, host, group=group)
        out.copy_(tmp)
        return
    dist.all_gather_into_tensor(out, mine, group=group)


def _all_gather_flat(mine: torch.Tensor, world: int, group=None) -> torch.Tensor:
    out = torch.empty(world * mine.numel(), dtype=mine.dtype, device=mine.device)
    _all_gather_into(out, mine, group)
    return out


def owner_of_model(model: int, n_models: int, world: int) -> int:
    """Rank that owns `model` under the contiguous block partition of ShardedDB."""
    for r in range(world):
        if (r * n_models) // world <= model < ((r + 1) * n_models) // world:
            return r
    raise ValueError(model)


class _DevMem:
    """__cuda_array_interface__ view over memory owned by the HIP library."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}


def _wrap_int32(ptr, n, dev):
    return torch.as_tensor(_DevMem(ptr, (n,), "<i4"), device=dev)


def _wrap_uint8(ptr, n, dev):
    return torch.as_tensor(_DevMem(ptr, (n,), "|u1"), device=dev)
