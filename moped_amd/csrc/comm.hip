// N > 1: the model database sharded over ranks, the two exchanges of a frame done inside the
// library.  A frame is
//     normalise + this shard's top-2   (mh_frame_enqueue_match_local)
//  -> all-gather of every shard's [3][Q] words + the previous frame's result block riding behind them
//  -> merge, CLUSTER .. FILTER2 on the models this rank owns   (mh_frame_enqueue_rest_batch)
// all on the context's stream, no host synchronisation.  The transport is RCCL (ncclAllGather over xGMI) --
// resolved at run time, so that a host without N > 1 never loads the 0.5 GB library -- or, for hosts whose
// ranks share a device or that bring their own transport (MPI, a test rig), a host callback.
//
// The caller this serves is the single-process frame loop of MopedPimpl::processImages
// (moped2/libmoped/src/moped.cpp:166-194): mh_comm_create_all + mh_frame_enqueue_sharded_all for a host that
// owns all devices, mh_comm_create + mh_frame_enqueue_sharded for one process per GPU.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "context.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  std::string err;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// The copy of RCCL already in the process (a torch host has its own) wins over a second one.
Rccl& rccl_state() {
  static Rccl r;
  return r;
}

void rccl_load(Rccl& r);

Rccl* rccl() {   // (thread safe: hosts with one thread per rank create their communicators concurrently)
  Rccl& r = rccl_state();
  static std::once_flag once;
  std::call_once(once, [&] { rccl_load(r); });
  return r.lib ? &r : nullptr;
}

void rccl_load(Rccl& r) {
  const char* env = std::getenv("MH_RCCL_PATH");
  if (env && *env) r.lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
  if (!r.lib && !(env && *env)) {
    for (const char* name : {"librccl.so.1", "librccl.so"})
      if (!r.lib) r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if (!r.lib) r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
  }
  if (!r.lib) {
    const char* e = dlerror();
    r.err = std::string("RCCL not loadable: ") + (e ? e : "?");
    return;
  }
  bool ok = true;
  auto sym = [&](const char* name) {
    void* p = dlsym(r.lib, name);
    if (!p) { ok = false; r.err = std::string("RCCL: missing symbol ") + name; }
    return p;
  };
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
  r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  if (!ok) r.lib = nullptr;
}

constexpr int EX2_WORDS = (16 + MH_EX2_OBJECTS * (int)sizeof(mh_object)) / 4;

}  // namespace

struct mh_comm {
  int rank = 0, world = 1, device = 0;
  ncclComm_t nccl = nullptr;
  mh_allgather_fn host_fn = nullptr;
  void* host_user = nullptr;
  std::vector<unsigned char> h_send, h_recv;
  uint32_t seq = 0;   // frame exchanges issued on this communicator: stamped into every block, compared after the gather
};

#define MH_NCCL(ctx, call)                                                                       \
  do {                                                                                           \
    ncclResult_t r_ = (call);                                                                    \
    if (r_ != ncclSuccess) {                                                                     \
      (ctx)->err = std::string(#call) + ": " + rccl()->GetErrorString(r_);                       \
      return MH_ERR_HIP;                                                                         \
    }                                                                                            \
  } while (0)

namespace {

// One all-gather of `bytes` (a multiple of 4) per rank on the context's stream.
int comm_allgather(mh_ctx* ctx, mh_comm* comm, const void* send_dev, void* recv_dev, size_t bytes) {
  if (comm->nccl) {
    MH_NCCL(ctx, rccl()->AllGather(send_dev, recv_dev, bytes / 4, ncclInt32, comm->nccl, ctx->stream));
    return MH_OK;
  }
  // host transport: the stream waits for the host here -- a rehearsal / portability path, not the fast one
  comm->h_send.resize(bytes);
  comm->h_recv.resize(bytes * (size_t)comm->world);
  MH_HIP(ctx, hipMemcpyAsync(comm->h_send.data(), send_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (comm->host_fn(comm->host_user, comm->h_send.data(), comm->h_recv.data(), bytes) != 0) {
    ctx->err = "mh_comm: the host all-gather callback failed";
    return MH_ERR_ARG;
  }
  MH_HIP(ctx, hipMemcpyAsync(recv_dev, comm->h_recv.data(), bytes * (size_t)comm->world, hipMemcpyHostToDevice,
                             ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));   // h_recv is reused by the next call
  return MH_OK;
}

// The context's exchange blocks: send = [3][BQ] top-2 words + B result heads, receive = world of those.
int ensure_exchange(mh_ctx* ctx, int world, int BQ, int B) {
  auto& ex = ctx->ex;
  const size_t stride = 3 * (size_t)BQ + (size_t)B * EX2_WORDS;
  if (ex.local && ex.stride == stride && ex.world == world && ex.batch == B) return MH_OK;
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));   // an earlier frame may still read the old blocks
  if (ex.cap_local < stride) {
    if (ex.local) hipFree(ex.local);
    ex.local = nullptr;
    ex.cap_local = 0;
    MH_HIP(ctx, hipMalloc(&ex.local, stride * 4));
    ex.cap_local = stride;
  }
  if (ex.cap_gather < stride * world) {
    if (ex.gathered) hipFree(ex.gathered);
    ex.gathered = nullptr;
    ex.cap_gather = 0;
    MH_HIP(ctx, hipMalloc(&ex.gathered, stride * world * 4));
    ex.cap_gather = stride * world;
  }
  MH_HIP(ctx, hipMemsetAsync(ex.local, 0, stride * 4, ctx->stream));
  MH_HIP(ctx, hipMemsetAsync(ex.gathered, 0, stride * world * 4, ctx->stream));
  ex.stride = stride;
  ex.world = world;
  ex.batch = B;
  ex.bq = BQ;
  return MH_OK;
}

int check_frame_args(mh_ctx* ctx, mh_comm* comm, const float* q_desc_dev, const float* q_uv_dev, int Q, int B,
                     const mh_cam* cam, const mh_frame_params* prm) {
  if (!ctx || !comm || !q_desc_dev || !q_uv_dev || Q <= 0 || B < 1 || B > MH_MAX_BATCH || !cam || !prm)
    return MH_ERR_ARG;
  if (comm->device != ctx->device) {
    ctx->err = "mh_comm: the communicator was made for another device";
    return MH_ERR_ARG;
  }
  if (B > 1 && (ctx->q_depth || ctx->depth_img.img || ctx->rules.on)) {
    ctx->err = "sharded batch: depth attributes and depth maps / rules belong to ONE frame";
    return MH_ERR_ARG;
  }
  return MH_OK;
}

// exchange 2 of the context's previous frame(s) + this shard's top-2 into the send block
int before_exchange(mh_ctx* ctx, mh_comm* comm, float* q_desc_dev, int Q, int B, const uint64_t* seeds) {
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc = mh::use_stream(ctx)) return rc;
  if (int rc = ensure_exchange(ctx, comm->world, B * Q, B)) return rc;
  if (int rc = mh_frame_result_copy_slots_dev(ctx, ctx->ex.local + 3 * (size_t)B * Q, B, MH_EX2_OBJECTS)) return rc;
  // the block's tag, in the two pad words of the first result head {n, flags, pad, pad}: which exchange of this
  // communicator this is, and for which frames -- every rank's must agree after the gather (group_kernel checks)
  int32_t* tag = ctx->ex.local + 3 * (size_t)B * Q + 2;
  const uint32_t fold = (uint32_t)seeds[0] ^ (uint32_t)(seeds[0] >> 32) ^ ((uint32_t)B << 24) ^ (uint32_t)Q;
  MH_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)tag, (int)comm->seq, 1, ctx->stream));
  MH_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)(tag + 1), (int)fold, 1, ctx->stream));
  ++comm->seq;
  void* block = nullptr;
  int64_t bytes = 0;
  if (mh_frame_result_dev(ctx, &block, &bytes) == MH_OK && bytes < EX2_WORDS * 4) {
    ctx->err = "sharded frame: the context's result block holds fewer than MH_EX2_OBJECTS objects";
    return MH_ERR_CAPACITY;
  }
  return mh_frame_enqueue_match_local(ctx, q_desc_dev, B * Q, ctx->ex.local);
}

int after_exchange(mh_ctx* ctx, mh_comm* comm, const float* q_uv_dev, int Q, int B, const mh_cam* cam,
                   const mh_frame_params* prm, const uint64_t* seeds) {
  ctx->exchange_tags = ctx->ex.gathered + 3 * (size_t)B * Q + 2;
  const int rc = mh_frame_enqueue_rest_frames(ctx, q_uv_dev, Q, ctx->ex.gathered, comm->world, (int)ctx->ex.stride, B * Q, B,
                                              cam, prm, seeds);
  ctx->exchange_tags = nullptr;
  return rc;
}

// {n, flags, pad, pad, objects} -> objects_host; more objects than the block carries or capacity flags: an error,
// not a shortened list
int unpack_head(mh_ctx* ctx, const int32_t* head, int head_objects, mh_object* objects_host, int cap, int32_t* n_io) {
  const int n = head[0], flags = head[1];
  if (n < 0 || n > head_objects || flags != 0) {
    ctx->err = "exchange 2: a rank reported " + std::to_string(n) + " objects (the block holds " +
               std::to_string(head_objects) + "), capacity flags " + std::to_string(flags);
    return MH_ERR_CAPACITY;
  }
  const int room = cap - *n_io;
  const int take = n < room ? n : (room > 0 ? room : 0);
  if (take > 0 && objects_host) std::memcpy(objects_host + *n_io, head + 4, sizeof(mh_object) * (size_t)take);
  *n_io += n;
  return MH_OK;
}

// mh_frame_fetch_previous_async: frame f of the batch BEFORE the current one -- every rank's head behind its top-2 words in
// the gathered exchange block -> one record of the caller's block: the ranks' objects one after the other in rank order.
constexpr int DELIVER_MAX_WORLD = 64;
__global__ void __launch_bounds__(128) deliver_previous_kernel(const int32_t* __restrict__ gathered, size_t stride_words, int bq,
                                                               int world, unsigned char* __restrict__ dst, int max_objects,
                                                               uint32_t tag) {
  __shared__ int base[DELIVER_MAX_WORLD + 1];
  const int f = blockIdx.x, t = threadIdx.x;
  const int32_t* head0 = gathered + 3 * (size_t)bq + (size_t)f * EX2_WORDS;
  if (t == 0) {
    int tot = 0, carried = 0, flags = 0;
    for (int r = 0; r < world; ++r) {
      const int32_t* h = head0 + (size_t)r * stride_words;
      const int n = h[0] > 0 ? h[0] : 0;
      flags |= h[1];
      if (n > MH_EX2_OBJECTS) flags |= 1 << 30;
      base[r] = carried;
      carried += n < MH_EX2_OBJECTS ? n : MH_EX2_OBJECTS;
      tot += n;
    }
    base[world] = carried;
    int32_t* out = reinterpret_cast<int32_t*>(dst + (size_t)f * (sizeof(mh_frame_head) + sizeof(mh_object) * (size_t)max_objects));
    out[0] = tot;
    out[1] = flags;
    out[2] = out[3] = out[4] = -1;
    out[5] = tot;
    out[6] = (int32_t)tag;
    out[7] = f;
  }
  __syncthreads();
  constexpr int OW = (int)(sizeof(mh_object) / 4);
  int32_t* out = reinterpret_cast<int32_t*>(dst + (size_t)f * (sizeof(mh_frame_head) + sizeof(mh_object) * (size_t)max_objects)) + 8;
  for (int r = 0; r < world; ++r) {
    const int32_t* h = head0 + (size_t)r * stride_words;
    const int first = base[r];
    const int take = min(base[r + 1], max_objects) - first;   // objects of rank r that fit
    for (int w = t; w < take * OW; w += blockDim.x) out[first * OW + w] = h[4 + w];
  }
  __threadfence_system();
}

}  // namespace

namespace mh {
void free_exchange(mh_ctx* ctx) {
  if (ctx->ex.local) hipFree(ctx->ex.local);
  if (ctx->ex.gathered) hipFree(ctx->ex.gathered);
  if (ctx->ex.flush) hipFree(ctx->ex.flush);
  ctx->ex = mh_ctx::Exchange();
}
}  // namespace mh

extern "C" {

int mh_comm_unique_id(unsigned char id[MH_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == MH_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  Rccl* r = rccl();
  if (!r || !id) return MH_ERR_ARG;
  ncclUniqueId u;
  if (r->GetUniqueId(&u) != ncclSuccess) return MH_ERR_HIP;
  std::memcpy(id, &u, sizeof u);
  return MH_OK;
}

int mh_comm_create(mh_ctx* ctx, const unsigned char id[MH_COMM_ID_BYTES], int rank, int world, mh_comm** out) {
  if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return MH_ERR_ARG;
  Rccl* r = rccl();
  if (!r) {
    ctx->err = "mh_comm_create: " + rccl_state().err + " (MH_RCCL_PATH names another copy)";
    return MH_ERR_ARG;
  }
  MH_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  ncclComm_t c = nullptr;
  MH_NCCL(ctx, r->CommInitRank(&c, world, u, rank));
  mh_comm* comm = new mh_comm;
  comm->rank = rank;
  comm->world = world;
  comm->device = ctx->device;
  comm->nccl = c;
  *out = comm;
  return MH_OK;
}

int mh_comm_create_all(mh_ctx* const* ctxs, int world, mh_comm** out) {
  if (!ctxs || !out || world < 1) return MH_ERR_ARG;
  for (int i = 0; i < world; ++i)
    if (!ctxs[i]) return MH_ERR_ARG;
  mh_ctx* ctx = ctxs[0];
  Rccl* r = rccl();
  if (!r) {
    ctx->err = "mh_comm_create_all: " + rccl_state().err + " (MH_RCCL_PATH names another copy)";
    return MH_ERR_ARG;
  }
  std::vector<int> devs(world);
  for (int i = 0; i < world; ++i) {
    devs[i] = ctxs[i]->device;
    for (int j = 0; j < i; ++j)
      if (devs[j] == devs[i]) {
        ctx->err = "mh_comm_create_all: two contexts on one device (one rank per device)";
        return MH_ERR_ARG;
      }
  }
  std::vector<ncclComm_t> comms(world, nullptr);
  MH_NCCL(ctx, r->CommInitAll(comms.data(), world, devs.data()));
  for (int i = 0; i < world; ++i) {
    mh_comm* comm = new mh_comm;
    comm->rank = i;
    comm->world = world;
    comm->device = devs[i];
    comm->nccl = comms[i];
    out[i] = comm;
  }
  return MH_OK;
}

int mh_comm_create_host(mh_ctx* ctx, int rank, int world, mh_allgather_fn fn, void* user, mh_comm** out) {
  if (!ctx || !out || !fn || world < 1 || rank < 0 || rank >= world) return MH_ERR_ARG;
  mh_comm* comm = new mh_comm;
  comm->rank = rank;
  comm->world = world;
  comm->device = ctx->device;
  comm->host_fn = fn;
  comm->host_user = user;
  *out = comm;
  return MH_OK;
}

int mh_comm_destroy(mh_comm* comm) {
  if (!comm) return MH_OK;
  int rc = MH_OK;
  if (comm->nccl) {
    hipSetDevice(comm->device);
    if (rccl()->CommDestroy(comm->nccl) != ncclSuccess) rc = MH_ERR_HIP;
  }
  delete comm;
  return rc;
}

int mh_comm_info(const mh_comm* comm, int* rank, int* world, int* is_rccl) {
  if (!comm) return MH_ERR_ARG;
  if (rank) *rank = comm->rank;
  if (world) *world = comm->world;
  if (is_rccl) *is_rccl = comm->nccl != nullptr;
  return MH_OK;
}

int mh_frame_enqueue_sharded_batch(mh_ctx* ctx, mh_comm* comm, float* q_desc_dev, const float* q_uv_dev, int Q, int B,
                                   const mh_cam* cam, const mh_frame_params* prm, const uint64_t* seeds) {
  if (!seeds) return MH_ERR_ARG;
  if (int rc = check_frame_args(ctx, comm, q_desc_dev, q_uv_dev, Q, B, cam, prm)) return rc;
  if (int rc = before_exchange(ctx, comm, q_desc_dev, Q, B, seeds)) return rc;
  if (int rc = comm_allgather(ctx, comm, ctx->ex.local, ctx->ex.gathered, ctx->ex.stride * 4)) return rc;
  return after_exchange(ctx, comm, q_uv_dev, Q, B, cam, prm, seeds);
}

int mh_frame_enqueue_sharded(mh_ctx* ctx, mh_comm* comm, float* q_desc_dev, const float* q_uv_dev, int Q,
                             const mh_cam* cam, const mh_frame_params* prm, uint64_t seed) {
  return mh_frame_enqueue_sharded_batch(ctx, comm, q_desc_dev, q_uv_dev, Q, 1, cam, prm, &seed);
}

int mh_frame_enqueue_sharded_all(mh_ctx* const* ctxs, mh_comm* const* comms, int world, float* const* q_desc_dev,
                                 const float* const* q_uv_dev, int Q, int B, const mh_cam* cam,
                                 const mh_frame_params* prm, const uint64_t* seeds) {
  if (!ctxs || !comms || !q_desc_dev || !q_uv_dev || !seeds || world < 1) return MH_ERR_ARG;
  for (int r = 0; r < world; ++r) {
    if (!ctxs[r] || !comms[r]) return MH_ERR_ARG;
    if (int rc = check_frame_args(ctxs[r], comms[r], q_desc_dev[r], q_uv_dev[r], Q, B, cam, prm)) return rc;
    if (!comms[r]->nccl || comms[r]->world != world || comms[r]->rank != r) {
      ctxs[r]->err = "mh_frame_enqueue_sharded_all: comms[r] must be rank r of a mh_comm_create_all set";
      return MH_ERR_ARG;
    }
  }
  for (int r = 0; r < world; ++r)
    if (int rc = before_exchange(ctxs[r], comms[r], q_desc_dev[r], Q, B, seeds)) return rc;
  // one thread driving several ranks: the collectives of all of them go out as one group
  mh_ctx* c0 = ctxs[0];
  MH_NCCL(c0, rccl()->GroupStart());
  for (int r = 0; r < world; ++r) {
    const ncclResult_t e = rccl()->AllGather(ctxs[r]->ex.local, ctxs[r]->ex.gathered, ctxs[r]->ex.stride, ncclInt32,
                                             comms[r]->nccl, ctxs[r]->stream);
    if (e != ncclSuccess) {
      rccl()->GroupEnd();
      c0->err = std::string("ncclAllGather: ") + rccl()->GetErrorString(e);
      return MH_ERR_HIP;
    }
  }
  MH_NCCL(c0, rccl()->GroupEnd());
  for (int r = 0; r < world; ++r)
    if (int rc = after_exchange(ctxs[r], comms[r], q_uv_dev[r], Q, B, cam, prm, seeds)) return rc;
  return MH_OK;
}

int mh_frame_previous_objects(mh_ctx* ctx, int frame_in_batch, mh_object* objects_host, int cap, int32_t* n_objects) {
  if (!ctx || !n_objects || cap < 0 || !ctx->ex.gathered || frame_in_batch < 0 || frame_in_batch >= ctx->ex.batch)
    return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  auto& ex = ctx->ex;
  ex.host.resize((size_t)ex.world * EX2_WORDS);
  MH_HIP(ctx, hipMemcpy2DAsync(ex.host.data(), EX2_WORDS * 4,
                               ex.gathered + 3 * (size_t)ex.bq + (size_t)frame_in_batch * EX2_WORDS, ex.stride * 4,
                               EX2_WORDS * 4, ex.world, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *n_objects = 0;
  for (int r = 0; r < ex.world; ++r)
    if (int rc = unpack_head(ctx, ex.host.data() + (size_t)r * EX2_WORDS, MH_EX2_OBJECTS, objects_host, cap, n_objects))
      return rc;
  return MH_OK;
}

int mh_frame_fetch_previous_async(mh_ctx* ctx, int max_objects, void* host_block, uint32_t tag) {
  if (!ctx || !host_block || max_objects < 0 || !ctx->ex.gathered || ctx->ex.batch < 1) return MH_ERR_ARG;
  auto& ex = ctx->ex;
  if (ex.world > DELIVER_MAX_WORLD) {
    ctx->err = "mh_frame_fetch_previous_async: more than 64 ranks";
    return MH_ERR_CAPACITY;
  }
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc = mh::use_stream(ctx)) return rc;
  const size_t bytes = mh_frame_block_stride(max_objects) * (size_t)ex.batch;
  unsigned char* dst = nullptr;
  if (int rc = mh::delivery_begin(ctx, host_block, bytes, &dst)) return rc;
  hipLaunchKernelGGL(deliver_previous_kernel, dim3(ex.batch), dim3(128), 0, ctx->stream, ex.gathered, ex.stride, ex.bq, ex.world,
                     dst, max_objects, tag);
  return mh::delivery_end(ctx, host_block, bytes, dst, ex.batch, max_objects, tag);
}

int mh_frame_gather_objects(mh_ctx* ctx, mh_comm* comm, int slot, mh_object* objects_host, int cap,
                            int32_t* n_objects) {
  if (!ctx || !comm || !n_objects || cap < 0 || slot < 0 || slot >= MH_MAX_BATCH) return MH_ERR_ARG;
  MH_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc = mh::use_stream(ctx)) return rc;
  void* block = nullptr;
  int64_t bytes = 0;
  if (int rc = mh_frame_result_dev(ctx, &block, &bytes)) return rc;
  auto& ex = ctx->ex;
  const size_t need = (size_t)bytes * comm->world;
  if (ex.flush_bytes < need) {
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ex.flush) hipFree(ex.flush);
    ex.flush = nullptr;
    ex.flush_bytes = 0;
    MH_HIP(ctx, hipMalloc(&ex.flush, need));
    ex.flush_bytes = need;
  }
  const unsigned char* mine = static_cast<const unsigned char*>(block) + (size_t)slot * bytes;
  if (int rc = comm_allgather(ctx, comm, mine, ex.flush, (size_t)bytes)) return rc;
  ex.host.resize(need / 4);
  MH_HIP(ctx, hipMemcpyAsync(ex.host.data(), ex.flush, need, hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *n_objects = 0;
  const int head_objects = (int)((bytes - 16) / (int64_t)sizeof(mh_object));
  for (int r = 0; r < comm->world; ++r)
    if (int rc = unpack_head(ctx, ex.host.data() + (size_t)r * (bytes / 4), head_objects, objects_host, cap, n_objects))
      return rc;
  return MH_OK;
}

}  // extern "C"
