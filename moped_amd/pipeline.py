"""Frame driver over the C ABI: device-resident MATCH -> CLUSTER -> POSE -> FILTER ->
POSE2 -> FILTER2, one mh_ctx per frame in flight, optional model sharding over
ranks with the two small exchanges of SURVEY.md 8(e).

PyTorch is used for what the C ABI does not own: device buffers for the frame
inputs, HIP streams, and handing rank 0's communicator id to the other ranks.  All
computation AND the exchanges (ncclAllGather on the slot's stream, csrc/comm.hip)
happen inside libmoped_hip.so.
"""
from __future__ import annotations

import os as _os
# one hardware queue per frame slot: ROCm's default of 4 caps the overlap of the slots' streams at four kernels (image ->
# objects: 1 540 -> 2 920 frames/s with 16).  Read by the HIP runtime when it creates its queues, so this only helps if the
# process has not touched the GPU yet; a C++ host exports it before it starts.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np
import torch

from . import capi


def models_of_rank(n_models: int, rank: int, world: int, assign: str = "block"):
    """Models rank `rank` of `world` owns: contiguous blocks [r n/W, (r+1) n/W), or round-robin (m % W == r --
    SURVEY.md 8(e): interleaving spreads the visible models' CLUSTER / POSE work over the ranks)."""
    if assign == "round-robin":
        return np.arange(rank, n_models, world, dtype=np.int64)
    if assign != "block":
        raise ValueError(f"unknown model assignment {assign!r}")
    return np.arange((rank * n_models) // world, ((rank + 1) * n_models) // world, dtype=np.int64)


class ShardedDB:
    """This rank's slice of the model database.  Rows of a model are contiguous in the flattened DB
    (MATCH_ANN_CPU::Update order), so a shard is a list of runs of global rows, in ascending order: ONE run for the
    block assignment (`row_lo` = mh_db_upload's index_base), one per maximal run of owned models for round-robin
    (mh_db_upload_blocks).  Row and model ids stay global either way."""

    def __init__(self, desc, xyz, model_of, n_models, rank=0, world=1, assign="block"):
        model_of = np.asarray(model_of, np.int32)
        owned = np.zeros(n_models + 1, bool)
        owned[models_of_rank(n_models, rank, world, assign)] = True
        mine = owned[model_of]
        rows = np.nonzero(mine)[0]
        # maximal runs of consecutive global rows
        starts = rows[np.r_[True, np.diff(rows) != 1]] if len(rows) else np.zeros(0, np.int64)
        ends = rows[np.r_[np.diff(rows) != 1, True]] + 1 if len(rows) else np.zeros(0, np.int64)
        self.block_global_row = starts.astype(np.int32)
        self.block_rows = (ends - starts).astype(np.int32)
        self.rows = rows.astype(np.int32)   # global row of every local row
        self.row_lo = int(rows[0]) if len(rows) else 0
        self.row_hi = int(rows[-1]) + 1 if len(rows) else 0
        self.desc = np.ascontiguousarray(desc[rows], np.float32)
        self.xyz = np.ascontiguousarray(xyz[rows], np.float32)
        self.model_of = np.ascontiguousarray(model_of[rows])
        self.n_models = n_models          # model ids stay global
        self.rank, self.world, self.assign = rank, world, assign

    def upload(self, ctx: "capi.Context", normalized):
        if len(self.block_rows) > 1:
            ctx.db_upload_blocks(normalized, self.model_of, self.xyz, self.n_models, self.block_global_row, self.block_rows)
        else:
            ctx.db_upload(normalized, self.model_of, self.xyz, self.n_models, index_base=self.row_lo)


class FramePipeline:
    """`depth` frames in flight on one GPU; each has its own mh_ctx + HIP stream so
    the latency-bound CLUSTER/POSE/FILTER kernels of one frame overlap the MATCH
    kernel of the next.

    With a sharded DB (world > 1, or force_exchange to run the same code on one rank) every frame goes through
    mh_frame_enqueue_sharded[_batch]: the exchanges happen inside libmoped_hip.so, on the slot's stream, over
    communicators this class only creates --
      * RCCL (mh_comm_create) when torch.distributed runs on nccl or is not initialised (world 1): rank 0's id
        travels by broadcast_object_list; `n_comms` communicators, slot i uses communicator i % n_comms;
      * the host transport (mh_comm_create_host over the gloo group) when the group is gloo: ranks that share one
        device (RCCL refuses that), the rehearsal of tests/test_gpu_dist.py."""

    def __init__(self, device: int, db: ShardedDB, depth: int = 1, max_queries: int = 4096,
                 params: capi.mh_frame_params | None = None, K=None, cam=None, group=None,
                 force_exchange: bool = False, n_comms: int = 4, batch: int = 1, lane: "tuple | None" = None,
                 id_leader: "int | None" = None):
        from . import synth
        self.dev = torch.device(f"cuda:{device}")
        torch.cuda.set_device(self.dev)
        self.db = db
        self.params = params or capi.default_frame_params()
        self.K = synth.K_DEFAULT if K is None else K
        self.cam = synth.CAM_IDENTITY if cam is None else cam
        self.group = group
        # A models x frames grid (bench.py --parallelism grid): this pipeline's communicators span the db.world ranks of ONE
        # frame group.  `id_leader` = the rank (of the default process group) that is shard 0 of this rank's frame group: the
        # RCCL ids then travel by ONE all_gather_object over the default group per communicator, which every rank of the
        # job joins -- no torch-side NCCL communicator per frame group.  `group` (a gloo subgroup of the frame group's ranks)
        # carries the host transport of the rehearsal instead.
        self.id_leader = id_leader
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
                db.upload(c, normalized)
            else:
                c.db_share(self.ctxs[0])   # one store per GPU: every frame in flight searches the same copy
            if batch > 1:   # frames travel in batches: per-frame working arrays, `batch` copies (mh_reserve_batch)
                c.reserve_batch(-(-max_queries // batch), batch)
            else:
                c.reserve(max_queries)
            self.ctxs.append(c)
            self.streams.append(s)
        self.depth = depth
        # lane = (n_streams, reserve_cus_per_xcd, low_priority): the contexts' chip-filling MATCH passes on shared streams
        self.lane = None
        if lane:
            self.lane = capi.Lane(device, *lane)
            for c in self.ctxs:
                c.set_lane(self.lane)
        self.exchange = force_exchange or self.world > 1
        self.comms = []
        if self.exchange:
            self.comms = self._make_comms(max(1, min(n_comms, depth)))
        self._batch = [1] * depth
        self._inputs = [None] * depth   # the slot's current input tensors: its stream may still be reading them
        self._cam = capi.make_cam(self.K, self.cam)

    def _make_comms(self, n):
        import torch.distributed as dist
        inited = dist.is_available() and dist.is_initialized()
        if inited and dist.get_backend(self.group) == "gloo":
            def allgather(blob: bytes) -> bytes:
                mine = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
                out = torch.empty(self.world * mine.numel(), dtype=torch.uint8)
                dist.all_gather_into_tensor(out, mine, group=self.group)
                return out.numpy().tobytes()
            return [capi.Comm.create_host(self.ctxs[0], self.db.rank, self.world, allgather)]
        comms = []
        for _ in range(n):
            ids = [capi.comm_unique_id() if self.db.rank == 0 else None]
            if self.id_leader is not None and inited:
                every = [None] * dist.get_world_size()
                dist.all_gather_object(every, ids[0])
                ids[0] = every[self.id_leader]
            elif self.world > 1:
                src = 0 if self.group is None else dist.get_global_rank(self.group, 0)
                dist.broadcast_object_list(ids, src=src, group=self.group)
            comms.append(capi.Comm.create(self.ctxs[0], ids[0], self.db.rank, self.world))
        return comms

    def _comm(self, slot):
        return self.comms[slot % len(self.comms)]

    def comm_info(self):
        """What carries the frames' exchange: (rank, world, transport) of this rank's communicators."""
        if not self.comms:
            return None
        r, w, rccl = self.comms[0].info()
        return {"rank": r, "world": w, "transport": "RCCL ncclAllGather" if rccl else "host callback (gloo)",
                "communicators": len(self.comms)}

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
        self._batch[slot] = 1
        if not self.exchange:
            c.frame_enqueue(q_desc.data_ptr(), q_uv.data_ptr(), Q, self.K, self.cam, self.params, seed)
            return
        c.frame_enqueue_sharded(self._comm(slot), q_desc.data_ptr(), q_uv.data_ptr(), Q, self.K, self.cam, self.params,
                                seed, _cam_struct=self._cam)

    # ---- batches of frames (small shards) ----------------------------------------------------
    def enqueue_batch(self, slot: int, q_desc: torch.Tensor, q_uv: torch.Tensor, B: int, seeds):
        """B frames through ONE MATCH launch and ONE exchange (a shard of a few thousand rows does not
        fill the chip for the 3000 queries of one frame): q_desc [B*Q,128], q_uv [B*Q,2], the frames one
        after the other; their CLUSTER..FILTER2 run one after the other on the slot's stream and leave
        their objects in result slots 0..B-1.  With a sharded DB through the exchange, otherwise
        mh_frame_enqueue_batch."""
        assert 1 <= B <= capi.MAX_BATCH
        c = self.ctxs[slot]
        self._inputs[slot] = (q_desc, q_uv)
        self._batch[slot] = B
        Q = q_desc.shape[0] // B
        if not self.exchange:
            c.frame_enqueue_batch(q_desc.data_ptr(), q_uv.data_ptr(), Q, B, self.K, self.cam, self.params, seeds,
                                  _cam_struct=self._cam)
            return
        c.frame_enqueue_sharded_batch(self._comm(slot), q_desc.data_ptr(), q_uv.data_ptr(), Q, B, self.K, self.cam,
                                      self.params, seeds, _cam_struct=self._cam)

    def fetch_batch(self, slot: int, B: int):
        return [self.ctxs[slot].frame_fetch_slot(f) for f in range(B)]

    # ---- delivery: every batch's objects into pinned host memory, no stream synchronisation ---------------
    def attach_delivery(self, max_objects: int = 16, B: int = capi.MAX_BATCH):
        """One pinned host block per slot for mh_frame_fetch_batch_async / _previous_async: records of
        mh_frame_head + mh_object[max_objects] for up to B frames."""
        self._dlv_cap = max_objects
        self._dlv_dtype = capi.frame_block_dtype(max_objects)
        self._dlv_host = [torch.zeros(capi.frame_block_bytes(B, max_objects), dtype=torch.uint8).pin_memory()
                          for _ in range(self.depth)]
        self._dlv_view = [t.numpy().view(self._dlv_dtype) for t in self._dlv_host]
        self._dlv_pending = [0] * self.depth   # frames of the slot's delivery in flight

    def deliver(self, slot: int, tag: int = 0):
        """Behind the batch just enqueued in `slot`: its objects (single GPU) or, with a sharded DB, all ranks' objects of
        the slot's PREVIOUS batch (they arrived with this batch's exchange) into the slot's host block."""
        c, B = self.ctxs[slot], self._batch[slot]
        if self.exchange:
            c.frame_fetch_previous_async(self._dlv_cap, self._dlv_host[slot].data_ptr(), tag)
        else:
            c.frame_fetch_batch_async(B, self._dlv_cap, self._dlv_host[slot].data_ptr(), tag)
        self._dlv_pending[slot] = B

    def take_delivery(self, slot: int):
        """Waits for the slot's delivery; the B records (a view of the pinned block: copy what must outlive the slot's
        next delivery) or None if nothing was pending."""
        B = self._dlv_pending[slot]
        if not B:
            return None
        self.ctxs[slot].frame_fetch_wait()
        self._dlv_pending[slot] = 0
        return self._dlv_view[slot][:B]

    def previous_objects_batch(self, slot: int):
        """Objects of the B frames of the batch enqueued in `slot` BEFORE the current one, from all ranks."""
        return [self.ctxs[slot].frame_previous_objects(f) for f in range(self._batch[slot])]

    def flush_objects_batch(self, slot: int, B: int):
        """Exchange 2 for the LAST batch of a slot (nothing follows to carry it)."""
        return [self.ctxs[slot].frame_gather_objects(self._comm(slot), f) for f in range(B)]

    def previous_objects(self, slot: int):
        """Objects of the frame enqueued in `slot` BEFORE the current one, from all ranks, as they
        arrived with the current frame's exchange (no collective of its own).  Synchronises the slot."""
        return self.ctxs[slot].frame_previous_objects(0)

    def fetch(self, slot: int):
        """Objects of the frame in `slot` (model ids global), counts[4]."""
        return self.ctxs[slot].frame_fetch()

    def gather_objects(self, slot: int):
        """Exchange 2 on its own: every rank's result block -> all ranks; the merged object array
        (rank order = model order)."""
        return self.ctxs[slot].frame_gather_objects(self._comm(slot), 0)

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        self.synchronize()
        for m in self.comms:
            m.close()
        self.comms = []
        for c in self.ctxs:
            c.close()
        self.ctxs = []
        if self.lane is not None:
            self.lane.close()
            self.lane = None


def exchange_top2(local: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """Exchange 1: one fused all-gather of every rank's per-query
    local top-2.  local = [3][Q] int32 words (idx1, bits of d1, bits of d2);
    returns [W][3][Q], the block mh_frame_enqueue_rest takes.  The layout of the exchange on
    torch tensors, for the world-size > 1 CPU tests (gloo); on the GPU the same all-gather is
    issued by the library itself (csrc/comm.hip)."""
    import torch.distributed as dist
    Q = local.shape[-1] if local.dim() == 2 else local.numel() // 3
    out = torch.empty(world * 3 * Q, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous().view(-1), group=group)
    return out.view(world, 3, Q)


def owner_of_model(model: int, n_models: int, world: int, assign: str = "block") -> int:
    """Rank that owns `model` under ShardedDB's partition."""
    if assign == "round-robin":
        return model % world
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
