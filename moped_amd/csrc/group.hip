// Ratio test + grouping of accepted matches by model in ascending query order:
// the tail of MATCH_ANN_CPU::process
// (moped2/libmoped/src/match/MATCH_ANN_CPU.hpp:165-176), plus the small
// bookkeeping kernels between the steps of a device-resident frame.
#include "steps.h"

MH_TRACE_TU()

namespace mh {

namespace {

constexpr int GROUP_THREADS = 1024;
constexpr int GROUP_MAX_MODELS = 8192;  // LDS histogram
#ifdef GROUP_PROF   // phase timing build (EXTRA=-DGROUP_PROF): cycles of thread 0 per phase
__device__ unsigned long long g_group_prof[8];
#define GP_T(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); g_group_prof[k] += now_ - t_prof; t_prof = now_; } } while (0)
#else
#define GP_T(k) do { } while (0)
#endif
constexpr int GROUP_LDS_M = 2048;       // accepted matches whose (model, pixel) wait in LDS for the rank / representative scans

__device__ __forceinline__ bool accepted(int32_t idx, float d1, float d2, float ratio) {
  // squared distances, fp32 division, exactly `ds[0]/ds[1] < Ratio` (:165)
  return idx >= 0 && (__fdiv_rn(d1, d2) < ratio);
}

// MATCH_ADAPTIVE_FLANN_CPU::getRatio (moped3d/libmoped/src/match/MATCH_ADAPTIVE_FLANN_CPU.hpp:193-215);
// cp = (maxRatioDepth, minRatioDepth, ratioLow, ratioHigh) of the model.
__device__ __forceinline__ float ratio_at(float depth, const float4 cp, float max_depth) {
  if (depth > max_depth) return 0.f;
  if (depth < cp.x) {
    const float progress = __fdiv_rn(depth, cp.x);
    return __fadd_rn(cp.z, __fmul_rn(progress, __fsub_rn(cp.w, cp.z)));
  } else if (depth < cp.y) {
    return cp.w;
  } else if (depth < __fmul_rn(cp.y, 2.f)) {
    const float progress = __fdiv_rn(__fsub_rn(__fmul_rn(cp.y, 2.f), depth), cp.y);
    return __fmul_rn(progress, cp.w);
  }
  return 0.f;
}
// getAdjustedRatio (:361-376): Cauchy weight of the pixel's fill distance blends the ratio at the
// measured depth with the ratio at DefaultDepth (double where the reference's literals make it so).
__device__ __forceinline__ float adjusted_ratio(float depth, float fill, const float4 cp, const DepthRules& R) {
  const float wt = __fdiv_rn(fill, R.cauchy_scale);
  const float weight = (float)(1.0 / (1.0 + (double)__fmul_rn(wt, wt)));
  const float put = ratio_at(depth, cp, R.max_depth), def = ratio_at(R.default_depth, cp, R.max_depth);
  return (float)((double)__fmul_rn(weight, put) + (1.0 - (double)weight) * (double)def);
}

__global__ void accept_kernel(const int32_t* __restrict__ idx1, const float* __restrict__ d1,
                              const float* __restrict__ d2, int Q, float ratio,
                              int32_t* __restrict__ out_idx) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < Q) out_idx[q] = accepted(idx1[q], d1[q], d2[q], ratio) ? idx1[q] : -1;
}

// Merge of the shards' top-2 (exchange 1 of a model-sharded DB): lower index wins a tie
// on the best distance, second best = min over the other candidates.
__device__ __forceinline__ void merge_top2(float& b1, float& b2, int32_t& i1, float ob1, float ob2,
                                           int32_t oi1) {
  const bool take = (ob1 < b1) || (ob1 == b1 && (unsigned)oi1 < (unsigned)i1);
  const float lose1 = take ? b1 : ob1;
  b2 = fminf(lose1, fminf(b2, ob2));
  b1 = take ? ob1 : b1;
  i1 = take ? oi1 : i1;
}

// Single workgroup, first launch of a frame after MATCH.  (0) reset the frame's
// counters; with a sharded DB, merge the gathered [S][3][Q] top-2 blocks into
// idx1/d1/d2; (a) ordered compaction of the accepted queries whose winning
// row belongs to this shard, (b) per-model histogram + exclusive scan, (c) stable
// placement by model, (d) m_rep: first match with the same image coordinate (the
// key of FILTER's bestPoints map, FILTER_PROJECTION_CPU.hpp:89); clears FILTER's
// per-keypoint claim table for the matches of this frame.
__global__ __launch_bounds__(GROUP_THREADS) void group_kernel(
    const int32_t* __restrict__ gathered, int n_shards, int32_t* idx1, float* d1, float* d2,
    int Q, float ratio, const float* __restrict__ q_uv, const int32_t* __restrict__ db_model,
    const float* __restrict__ db_xyz, int N, RowMap rmap, int n_models, int max_m,
    int32_t* __restrict__ acc_q, int32_t* __restrict__ acc_model, int32_t* __restrict__ m_q,
    int32_t* __restrict__ m_model, mh_corr* __restrict__ m_corr, int32_t* __restrict__ m_rep,
    int32_t* __restrict__ model_off, const mh_depth* __restrict__ q_depth,
    mh_depth* __restrict__ m_depth, DepthImage dimg, FrameCounts* counts, int32_t* __restrict__ n_slots,
    unsigned long long* __restrict__ best, DepthRules rules, int shard_stride, int plane_stride, FrameBatch fbx,
    const int32_t* __restrict__ tags, DepthMaps maps) {
  MH_TRACE_SCOPE(mh::TK_GROUP);
  if (blockIdx.y) {   // frame of a batch: its slice of the top-2 arrays / keypoints, its copy of the working arrays
    const unsigned long long a = blockIdx.y * fbx.arena;
    const size_t q0 = (size_t)blockIdx.y * fbx.q;
    idx1 += q0; d1 += q0; d2 += q0; q_uv += 2 * q0;
    if (q_depth) q_depth += q0;       // (per-query depth attributes lie frame after frame like the queries)
    if (gathered) gathered += q0;   // (the frames' columns of every shard's [3][B Q] block)
    acc_q = frame_ptr(acc_q, a); acc_model = frame_ptr(acc_model, a); m_q = frame_ptr(m_q, a);
    m_model = frame_ptr(m_model, a); m_corr = frame_ptr(m_corr, a); m_rep = frame_ptr(m_rep, a);
    model_off = frame_ptr(model_off, a); m_depth = frame_ptr(m_depth, a); counts = frame_ptr(counts, a);
    n_slots = frame_ptr(n_slots, a); best = frame_ptr(best, a);
    if (dimg.img) {   // a depth map per frame (DepthMaps), the frame's rule buffers behind those of the frames before it
      dimg.img = maps.img[blockIdx.y];
      dimg.fill = maps.fill[blockIdx.y];
      const size_t P = (size_t)rules.pw * rules.ph;
      if (rules.keep1) rules.keep1 += q0;
      if (rules.inv_size) rules.inv_size += blockIdx.y * P;
      if (rules.cnt) rules.cnt += blockIdx.y * P * n_models;
    }
  }
  __shared__ int hist[GROUP_MAX_MODELS + 1];
  __shared__ int wave_cnt[GROUP_THREADS / 64];
  __shared__ int pass_cnt[2][GROUP_THREADS / 64];
  __shared__ int base_s;
  __shared__ __attribute__((aligned(16))) int model_s[GROUP_LDS_M];
  __shared__ __attribute__((aligned(16))) float2 uv_s[GROUP_LDS_M];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef GROUP_PROF
  unsigned long long t_prof = clock64();
#endif
  for (int m = tid; m <= n_models; m += GROUP_THREADS) hist[m] = 0;
  if (tid == 0) {
    base_s = 0;
    *counts = FrameCounts{};
    if (n_slots) *n_slots = 0;
    // every rank stamps its block with (sequence number of the collective on its communicator, frame seed): blocks that
    // do not agree were gathered by collectives the ranks issued in different orders -- the merge below would mix frames
    if (tags && blockIdx.y == 0) {
      bool same = true;
      for (int k = 1; k < n_shards; ++k)
        same &= tags[(size_t)k * shard_stride] == tags[0] && tags[(size_t)k * shard_stride + 1] == tags[1];
      if (!same) counts->error = ERR_EXCHANGE;
    }
  }
  if (gathered) {
    const float* gf = reinterpret_cast<const float*>(gathered);
    for (int q = tid; q < Q; q += GROUP_THREADS) {
      float b1 = __builtin_inff(), b2 = __builtin_inff();
      int32_t i1 = -1;
      for (int k = 0; k < n_shards; ++k) {
        const size_t o = (size_t)k * shard_stride + q;   // shard k's [3][Q] block (+ what rides behind it)
        const int32_t i = gathered[o];
        if (i >= 0) merge_top2(b1, b2, i1, gf[o + plane_stride], gf[o + 2 * (size_t)plane_stride], i);
      }
      idx1[q] = i1;
      d1[q] = b1;
      d2[q] = b2;
    }
    __threadfence_block();
  }
  __syncthreads();

  GP_T(0);
  // (a) ordered compaction, 1024 queries per pass; the acceptance tests of four passes (dependent global
  // loads: idx1 -> the row's model, d1, d2) are evaluated together so that their latencies overlap
  constexpr int PASSES = 4;
  int base = 0, pass_no = 0;
  for (int q00 = 0; q00 < Q; q00 += PASSES * GROUP_THREADS) {
    bool okv[PASSES];
    int modelv[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int q = q00 + ps * GROUP_THREADS + tid;
      bool ok = false;
      int model = 0;
      if (q < Q) {
        const int32_t gi = idx1[q];
        const int32_t li = gi >= 0 ? row_to_local(rmap, gi, N) : -1;
        if (li >= 0 && !(rules.keep1 && !rules.keep1[q])) {
          model = db_model[li];
          float rq = ratio;
          bool reachable = true;
          if (rules.ratio_table) {
            // the feature's pixel (:447-450; clamped to the map, the reference clamps to [0, width])
            int x = (int)q_uv[2 * q], y = (int)q_uv[2 * q + 1];
            x = x < 0 ? 0 : (x >= dimg.w ? dimg.w - 1 : x);
            y = y < 0 ? 0 : (y >= dimg.h ? dimg.h - 1 : y);
            const float depth = dimg.img[(size_t)y * dimg.w + x].z;
            reachable = !(depth > rules.max_depth);   // "Don't even bother searching" (:457-460)
            const float fill = dimg.fill ? dimg.fill[(size_t)y * dimg.w + x] : 0.f;
            rq = adjusted_ratio(depth, fill, rules.ratio_table[model], rules);
          }
          ok = reachable && accepted(gi, d1[q], d2[q], rq);
        }
      }
      okv[ps] = ok;
      modelv[ps] = model;
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int q = q00 + ps * GROUP_THREADS + tid;
      if (q00 + ps * GROUP_THREADS >= Q) break;   // uniform
      const bool ok = okv[ps];
      const int model = modelv[ps];
      // one barrier per pass: the wavefronts' counts alternate between two buffers, every thread sums them
      // itself and carries the running base in a register
      const unsigned long long bal = __ballot(ok);
      int* const cnt = pass_cnt[pass_no & 1];
      ++pass_no;
      if (lane == 0) cnt[wave] = __popcll(bal);
      __syncthreads();
      int before = base, tot = 0;
#pragma unroll
      for (int w = 0; w < GROUP_THREADS / 64; ++w) {
        const int c = cnt[w];
        if (w < wave) before += c;
        tot += c;
      }
      base += tot;
      const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
      if (ok && pos < max_m) {
        acc_q[pos] = q;
        acc_model[pos] = model;
        atomicAdd(&hist[model], 1);
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  int M = base;
  if (M > max_m) M = max_m;
  GP_T(1);

  // (a') DEPTHFILTER with ToFilter = 2 (moped3d DEPTHFILTER_CPU.hpp:212-249): per model, the
  // density of its matches over the image patches, dilated, must exceed Density where the
  // match sits.  Counts per (model, patch) in global scratch (zero before and after).
  if (rules.inv_size) {
    const int P = rules.pw * rules.ph;
    for (int i = tid; i < M; i += GROUP_THREADS) {
      const int q = acc_q[i];
      atomicAdd(&rules.cnt[(size_t)acc_model[i] * P + patch_of(q_uv[2 * q], q_uv[2 * q + 1], rules.patch, rules.pw, rules.ph)], 1);
    }
    __threadfence_block();
    __syncthreads();
    for (int i = tid; i < M; i += GROUP_THREADS) {
      const int q = acc_q[i];
      const int32_t* cm = rules.cnt + (size_t)acc_model[i] * P;
      const int p = patch_of(q_uv[2 * q], q_uv[2 * q + 1], rules.patch, rules.pw, rules.ph);
      const float v = dilated_by([&](int pp) { return density_replay(cm[pp], rules.inv_size[pp]); }, p, rules.pw, rules.ph);
      m_rep[i] = v > rules.filter2;   // scratch until (d)
    }
    __threadfence_block();
    __syncthreads();
    for (int i = tid; i < M; i += GROUP_THREADS) {
      const int q = acc_q[i];
      rules.cnt[(size_t)acc_model[i] * P + patch_of(q_uv[2 * q], q_uv[2 * q + 1], rules.patch, rules.pw, rules.ph)] = 0;
    }
    // ordered compaction of the accepted list, in place (destinations never pass the sources)
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < M; i0 += GROUP_THREADS) {
      const int i = i0 + tid;
      bool keep = false;
      int q = 0, model = 0;
      if (i < M) {
        q = acc_q[i];
        model = acc_model[i];
        keep = m_rep[i] != 0;
        if (!keep) atomicSub(&hist[model], 1);
      }
      const unsigned long long bal = __ballot(keep);
      if (lane == 0) wave_cnt[wave] = __popcll(bal);
      __syncthreads();
      int before = base_s;
      for (int w = 0; w < wave; ++w) before += wave_cnt[w];
      const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
      if (keep) {
        acc_q[pos] = q;
        acc_model[pos] = model;
      }
      __syncthreads();
      if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < GROUP_THREADS / 64; ++w) tot += wave_cnt[w];
        base_s += tot;
      }
      __syncthreads();
    }
    M = base_s;
  }

  GP_T(2);
  // (b) exclusive scan of the histogram (single thread: n_models is small)
  if (tid == 0) {
    int run = 0;
    for (int m = 0; m < n_models; ++m) {
      const int c = hist[m];
      hist[m] = run;
      model_off[m] = run;
      run += c;
    }
    hist[n_models] = run;
    model_off[n_models] = run;
    counts->n_matches = M;
  }
  __syncthreads();

  GP_T(3);
  // (c) stable placement: an entry's place = its model's offset + its rank among the EARLIER accepted entries of the same
  // model.  Frames of up to GROUP_LDS_M matches and GROUP_LDS_M models: the entries' numbers are first dropped into
  // their model's bucket (an LDS counter per model, any order), the rank is the number of smaller numbers in the bucket --
  // the sum of the squares of the models' counts in LDS reads instead of M^2 / 8 (ten visible objects, 1 600 matches:
  // group_kernel 94 -> ~25 us).  Larger frames scan all earlier entries (an LDS copy of the models up to GROUP_LDS_M
  // entries, global memory beyond).
  const bool in_lds = M <= GROUP_LDS_M;
  const bool buckets = in_lds && n_models <= GROUP_LDS_M;
  int* const fill = model_s;                              // [n_models] running places of the buckets (bucket path)
  int* const blist = reinterpret_cast<int*>(uv_s);        // [M] entry numbers, bucket after bucket (until the placement writes uv_s)
  if (buckets) {
    for (int m = tid; m < n_models; m += GROUP_THREADS) fill[m] = hist[m];
    __syncthreads();
    for (int i = tid; i < M; i += GROUP_THREADS) blist[atomicAdd(&fill[acc_model[i]], 1)] = i;
    __syncthreads();
  } else if (in_lds) {
    for (int i = tid; i < M; i += GROUP_THREADS) model_s[i] = acc_model[i];
    __syncthreads();
  }
  constexpr int PER_T = GROUP_LDS_M / GROUP_THREADS;      // entries per thread while the ranks wait for the barrier below
  int dst_keep[PER_T];
  if (buckets) {
#pragma unroll
    for (int k = 0; k < PER_T; ++k) {
      const int i = tid + k * GROUP_THREADS;
      dst_keep[k] = 0;
      if (i < M) {
        const int model = acc_model[i];
        const int b0 = hist[model], b1 = hist[model + 1];
        int rank = 0, e = b0;
        for (; e + 4 <= b1; e += 4)   // four reads in flight per step
          rank += (int)(blist[e] < i) + (int)(blist[e + 1] < i) + (int)(blist[e + 2] < i) + (int)(blist[e + 3] < i);
        for (; e < b1; ++e) rank += (int)(blist[e] < i);
        dst_keep[k] = b0 + rank;
      }
    }
    __syncthreads();   // every rank is known: the placement below overwrites the buckets (uv_s)
  }
  for (int i = tid; i < M; i += GROUP_THREADS) {
    const int model = acc_model[i];
    int dst;
    if (buckets) {
      dst = dst_keep[(i - tid) / GROUP_THREADS];
    } else {
      int rank = 0;
      if (in_lds) {
        const int4* m4 = reinterpret_cast<const int4*>(model_s);   // four entries per LDS read
        int j = 0;
        for (; j + 16 <= i; j += 16) {   // four reads in flight per step
          const int4 v0 = m4[(j >> 2)], v1 = m4[(j >> 2) + 1], v2 = m4[(j >> 2) + 2], v3 = m4[(j >> 2) + 3];
          rank += (v0.x == model) + (v0.y == model) + (v0.z == model) + (v0.w == model) + (v1.x == model) + (v1.y == model) +
                  (v1.z == model) + (v1.w == model) + (v2.x == model) + (v2.y == model) + (v2.z == model) + (v2.w == model) +
                  (v3.x == model) + (v3.y == model) + (v3.z == model) + (v3.w == model);
        }
        for (; j + 4 <= i; j += 4) {
          const int4 v = m4[j >> 2];
          rank += (v.x == model) + (v.y == model) + (v.z == model) + (v.w == model);
        }
        for (; j < i; ++j) rank += (model_s[j] == model);
      } else {
        for (int j = 0; j < i; ++j) rank += (acc_model[j] == model);
      }
      dst = hist[model] + rank;
    }
    const int q = acc_q[i];
    const int32_t li = row_to_local(rmap, idx1[q], N);   // (an accepted match: the row is this shard's)
    m_q[dst] = q;
    m_model[dst] = model;
    mh_corr c;
    c.u = q_uv[2 * q];
    c.v = q_uv[2 * q + 1];
    c.x = db_xyz[3 * (size_t)li];
    c.y = db_xyz[3 * (size_t)li + 1];
    c.z = db_xyz[3 * (size_t)li + 2];
    m_corr[dst] = c;
    if (in_lds) uv_s[dst] = make_float2(c.u, c.v);
    if (q_depth) {
      m_depth[dst] = q_depth[q];
    } else if (dimg.img) {
      // DEPTHMAP_PROP_CPU::process (moped3d/libmoped/src/depthprop/DEPTHMAP_PROP_CPU.hpp:101-134):
      // nearest pixel (truncation), no interpolation; fillDistance -1 without a distance map;
      // weight = getCauchyWeight(fillDistance) (POSE_..._DEPTH_CPU.hpp:194-197,351).  Pixels
      // outside the map (the reference reads out of bounds there) are clamped to the border.
      int ix = (int)c.u, iy = (int)c.v;
      ix = ix < 0 ? 0 : (ix >= dimg.w ? dimg.w - 1 : ix);
      iy = iy < 0 ? 0 : (iy >= dimg.h ? dimg.h - 1 : iy);
      const float4 px = dimg.img[(size_t)iy * dimg.w + ix];
      const float fd = dimg.fill ? dimg.fill[(size_t)iy * dimg.w + ix] : -1.f;
      const float factor = __fdiv_rn(fd, dimg.cauchy_scale);
      mh_depth md;
      md.wx = px.x;
      md.wy = px.y;
      md.wz = px.z;
      md.w = (float)(1.0 / (double)__fadd_rn(1.f, __fmul_rn(factor, factor)));
      m_depth[dst] = md;
    }
  }
  __threadfence_block();
  __syncthreads();

  GP_T(4);
  // (d) representative of each image coordinate: the first earlier entry with the same pixel = the smallest such number.
  // Up to GROUP_LDS_M matches: a hash of the pixel's bits leads to a chain of the entries that share it (LDS heads and
  // links, built with atomic exchanges in any order), the smallest equal one on the chain is the representative -- M
  // short chains instead of M^2 / 4 comparisons.
  if (in_lds) {
    int* const head = model_s;   // [GROUP_LDS_M] first entry of a hash value's chain (-1: none)
    int* const link = hist;      // [M] next entry of the chain (the histogram has done its work: model_off holds it)
    for (int h = tid; h < GROUP_LDS_M; h += GROUP_THREADS) head[h] = -1;
    __syncthreads();
    auto hash_of = [](float2 p) {   // equal pixels (==) hash alike: -0 + 0 = +0
      const unsigned a = __float_as_uint(p.x + 0.f), b = __float_as_uint(p.y + 0.f);
      return (int)(((a * 0x9E3779B1u) ^ (b * 0x85EBCA77u)) >> 21) & (GROUP_LDS_M - 1);
    };
    static_assert(GROUP_LDS_M == 2048, "the hash keeps 11 bits");
    for (int i = tid; i < M; i += GROUP_THREADS) link[i] = atomicExch(&head[hash_of(uv_s[i])], i);
    __syncthreads();
    for (int i = tid; i < M; i += GROUP_THREADS) {
      const float2 p = uv_s[i];
      int rep = i;
      for (int j = head[hash_of(p)]; j >= 0; j = link[j]) {
        const float2 o = uv_s[j];
        if (j < rep && o.x == p.x && o.y == p.y) rep = j;
      }
      m_rep[i] = rep;
      if (best) best[i] = 0ull;
    }
  } else {
    for (int i = tid; i < M; i += GROUP_THREADS) {
      int rep = i;
      const float u = m_corr[i].u, v = m_corr[i].v;
      for (int j = 0; j < i; ++j)
        if (m_corr[j].u == u && m_corr[j].v == v) {
          rep = j;
          break;
        }
      m_rep[i] = rep;
      if (best) best[i] = 0ull;
    }
  }
#ifdef MH_TRACE
  __syncthreads();   // (trace builds: thread 0 files the workgroup's record when the LAST wavefront is through)
#endif
  GP_T(5);
}

#ifdef GROUP_PROF
}  // namespace
extern "C" int mh_debug_group_prof(unsigned long long out[8], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_group_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_group_prof), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
namespace {
#endif
// m_rep for caller-provided match lists (per-step FILTER entry point).
// See launch_image_split (steps.h).  One workgroup; frames with several images are the exception, the work is
// O(matches x matches-of-a-model) and nothing here is tuned.
__global__ __launch_bounds__(GROUP_THREADS) void image_split_kernel(
    const mh_corr* __restrict__ m_corr, const int32_t* __restrict__ m_q, const int32_t* __restrict__ m_model,
    const int32_t* __restrict__ model_off, int n_models, const int32_t* __restrict__ q_img, int n_images,
    const FrameCounts* __restrict__ counts, int32_t* __restrict__ m_img, int32_t* __restrict__ m_rep,
    mh_corr* __restrict__ mi_corr, int32_t* __restrict__ mi_img, int32_t* __restrict__ off2) {
  const int tid = threadIdx.x;
  const int M = counts->n_matches;
  for (int i = tid; i < M; i += GROUP_THREADS) {
    int im = q_img[m_q[i]];
    im = im < 0 ? 0 : (im >= n_images ? n_images - 1 : im);
    m_img[i] = im;
  }
  __threadfence_block();
  __syncthreads();
  // representative of every (coord2D, image)
  for (int i = tid; i < M; i += GROUP_THREADS) {
    const float u = m_corr[i].u, v = m_corr[i].v;
    const int im = m_img[i];
    int rep = i;
    for (int j = 0; j < i; ++j)
      if (m_corr[j].u == u && m_corr[j].v == v && m_img[j] == im) {
        rep = j;
        break;
      }
    m_rep[i] = rep;
  }
  // first row of every (model, image) set
  for (int vm = tid; vm <= n_models * n_images; vm += GROUP_THREADS) {
    if (vm == n_models * n_images) {
      off2[vm] = M;
      continue;
    }
    const int m = vm / n_images, im = vm % n_images;
    const int b = model_off[m], e = model_off[m + 1];
    int before = 0;
    for (int j = b; j < e; ++j) before += m_img[j] < im;
    off2[vm] = b + before;
  }
  // stable placement inside the model's slice: by image, then in list (= query) order
  for (int i = tid; i < M; i += GROUP_THREADS) {
    const int m = m_model[i], im = m_img[i];
    const int b = model_off[m], e = model_off[m + 1];
    int dst = b;
    for (int j = b; j < e; ++j) dst += (m_img[j] < im) || (m_img[j] == im && j < i);
    mi_corr[dst] = m_corr[i];
    mi_img[dst] = im;
  }
}

__global__ void rep_kernel(const mh_corr* __restrict__ corr, const int32_t* __restrict__ img, int M,
                           int32_t* __restrict__ rep_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const float u = corr[i].u, v = corr[i].v;
  int rep = i;
  for (int j = 0; j < i; ++j)
    if (corr[j].u == u && corr[j].v == v && (!img || img[j] == img[i])) {
      rep = j;
      break;
    }
  rep_out[i] = rep;
}

}  // namespace

void launch_accept(const int32_t* idx1, const float* d1, const float* d2, int Q, float ratio,
                   int32_t* out_idx, hipStream_t s) {
  if (Q <= 0) return;
  hipLaunchKernelGGL(accept_kernel, dim3((Q + 255) / 256), dim3(256), 0, s, idx1, d1, d2, Q, ratio,
                     out_idx);
}

void launch_group(const int32_t* gathered, int n_shards, int32_t* idx1, float* d1, float* d2, int Q,
                  float ratio, const float* q_uv, const int32_t* db_model, const float* db_xyz, int N,
                  const RowMap& rmap, int n_models, int max_m, int32_t* acc_q, int32_t* acc_model,
                  int32_t* m_q, int32_t* m_model, mh_corr* m_corr, int32_t* m_rep,
                  int32_t* model_off, const mh_depth* q_depth, mh_depth* m_depth, const DepthImage& dimg,
                  FrameCounts* counts, int32_t* n_slots, unsigned long long* best, hipStream_t s,
                  const DepthRules& rules, int shard_stride, int plane_stride, const FrameBatch* batch, const int32_t* tags,
                  const DepthMaps* maps) {
  hipLaunchKernelGGL(group_kernel, dim3(1, batch ? batch->n : 1), dim3(GROUP_THREADS), 0, s, gathered, n_shards, idx1, d1, d2,
                     Q, ratio, q_uv, db_model, db_xyz, N, rmap, n_models, max_m, acc_q, acc_model,
                     m_q, m_model, m_corr, m_rep, model_off, q_depth, m_depth, dimg, counts, n_slots, best, rules,
                     shard_stride > 0 ? shard_stride : 3 * Q, plane_stride > 0 ? plane_stride : Q,
                     batch ? *batch : FrameBatch(), gathered ? tags : nullptr, (batch && maps) ? *maps : DepthMaps());
}

void launch_image_split(const mh_corr* m_corr, const int32_t* m_q, const int32_t* m_model, const int32_t* model_off,
                        int n_models, const int32_t* q_img, int n_images, const FrameCounts* counts, int32_t* m_img,
                        int32_t* m_rep, mh_corr* mi_corr, int32_t* mi_img, int32_t* off2, hipStream_t s) {
  hipLaunchKernelGGL(image_split_kernel, dim3(1), dim3(GROUP_THREADS), 0, s, m_corr, m_q, m_model, model_off, n_models,
                     q_img, n_images, counts, m_img, m_rep, mi_corr, mi_img, off2);
}

void launch_rep(const mh_corr* corr, int M, int32_t* rep, hipStream_t s, const int32_t* img) {
  if (M <= 0) return;
  hipLaunchKernelGGL(rep_kernel, dim3((M + 255) / 256), dim3(256), 0, s, corr, img, M, rep);
}

}  // namespace mh
