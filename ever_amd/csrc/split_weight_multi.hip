// One launch that re-splits EVERY convolution weight of the model into the bf16 planes its kernels read
// (forward and data-gradient layouts alike).  The planes change once per optimiser step; as one kernel per
// convolution and direction that was 164 launches of ~5 us per FarSeg-R50 step (0.85 ms + boundaries) for
// 0.7 GB of traffic.  The host builds the job table once (evk_conv2d_split_jobs; it sets job.arg[12] = workgroups
// for that job, ~evk_split_job_pairs / 2048, and a (job, block-in-job) map with one row per workgroup), keeps both
// in device memory and calls evk_conv2d_split_multi after each weight update.
#include "split_weight.hpp"
#include "igemm_common.hpp"

namespace evk {

// block_map[b] = (job, block index inside the job); job.arg[12] = blocks the host gave that job (so every workgroup
// has work: a (blocks, jobs) grid sized for the largest weight would launch ~80 K empty workgroups for the small ones)
// absmax != nullptr: the f16x2 planes; job.arg[11] = index of the job's weight in the absmax array (evk_absmax_multi)
__global__ __launch_bounds__(256) void split_weight_multi_kernel(const evk_split_job* __restrict__ jobs,
                                                                 const int32_t* __restrict__ block_map,
                                                                 const uint32_t* __restrict__ absmax) {
  const int jb = block_map[2 * blockIdx.x], bj = block_map[2 * blockIdx.x + 1];
  const evk_split_job j = jobs[jb];
  const size_t t0 = (size_t)bj * 256 + threadIdx.x, nt = (size_t)j.arg[12] * 256;
  const float* w = j.w;
  uint16_t* out = reinterpret_cast<uint16_t*>(j.out);
  const uint32_t* wscale = absmax ? absmax + j.arg[11] : nullptr;
  if (j.kind == kSplitFwd)
    split_fwd_body(w, out, j.arg[0], j.arg[1], j.arg[2], t0, nt, wscale);
  else if (j.kind == kSplitDgrad)
    split_dgrad_body(w, out, j.arg[0], j.arg[1], j.arg[2], j.arg[3], j.arg[4], j.arg[5], j.arg[6], j.arg[7], j.arg[8],
                     j.arg[9], j.arg[10], t0, nt, wscale);
  else if (wscale && j.arg[3])   // f16x2 and a shape the Winograd kernel takes (conv_desc_uses_wino)
    split_wino_body(w, out, j.arg[0], j.arg[1], j.arg[2], t0, nt, wscale);
  else
    split_halo_body(w, out, j.arg[0], j.arg[1], j.arg[2], t0, nt, wscale);
}

static inline int kpad32(int k) { return (k + 31) & ~31; }

}  // namespace evk

using namespace evk;

extern "C" int32_t evk_conv2d_split_job_count(const evk_conv_desc* d, int32_t for_dgrad) {
  if (!d || d->stride_h <= 0 || d->stride_w <= 0) return 0;
  if (d->kh == 3 && d->kw == 3 && conv_desc_uses_halo(d, for_dgrad ? 1 : 0)) return 1;
  return for_dgrad ? d->stride_h * d->stride_w : 1;
}

extern "C" int evk_conv2d_split_jobs(const evk_conv_desc* d, const float* w, int32_t for_dgrad, void* wsplit,
                                     evk_split_job* jobs, int32_t max_jobs) {
  EVK_REQUIRE(d && w && wsplit && jobs, EVK_E_INVALID, "split_jobs: null pointer");
  EVK_REQUIRE(d->stride_h > 0 && d->stride_w > 0 && d->dil_h > 0 && d->dil_w > 0, EVK_E_INVALID,
              "split_jobs: bad stride/dilation");
  EVK_REQUIRE(max_jobs >= evk_conv2d_split_job_count(d, for_dgrad), EVK_E_INVALID, "split_jobs: job array too small");
  uint16_t* out = reinterpret_cast<uint16_t*>(wsplit);
  int n = 0;
  auto put = [&](int kind, uint16_t* o) -> evk_split_job& {
    evk_split_job& j = jobs[n++];
    j.w = w; j.out = o; j.kind = kind;
    for (int i = 0; i < 13; ++i) j.arg[i] = 0;
    return j;
  };
  // the same layout decisions as evk_conv2d_split_weight (the consumer kernels make them from the descriptor too)
  if (d->kh == 3 && d->kw == 3 && conv_desc_uses_halo(d, for_dgrad ? 1 : 0)) {
    evk_split_job& j = put(kSplitHalo, out);
    j.arg[0] = d->Cout; j.arg[1] = d->Cin; j.arg[2] = for_dgrad ? 1 : 0;
    j.arg[3] = conv_desc_uses_wino(d, for_dgrad ? 1 : 0) ? 1 : 0;
    return n;
  }
  if (!for_dgrad) {
    const int K = d->kh * d->kw * d->Cin;
    evk_split_job& j = put(kSplitFwd, out);
    j.arg[0] = d->Cout; j.arg[1] = K; j.arg[2] = kpad32(K);
    return n;
  }
  size_t off = 0;
  for (int cy = 0; cy < d->stride_h; ++cy)
    for (int cx = 0; cx < d->stride_w; ++cx) {
      const AxisPlan py = plan_axis(cy, d->pad_h, d->dil_h, d->stride_h, d->kh);
      const AxisPlan px = plan_axis(cx, d->pad_w, d->dil_w, d->stride_w, d->kw);
      const int K = py.nt * px.nt * d->Cout, Kp = kpad32(K);
      if (K > 0) {
        evk_split_job& j = put(kSplitDgrad, out + off);
        j.arg[0] = d->Cout; j.arg[1] = d->kh; j.arg[2] = d->kw; j.arg[3] = d->Cin;
        j.arg[4] = py.k0; j.arg[5] = py.kstep; j.arg[6] = py.nt;
        j.arg[7] = px.k0; j.arg[8] = px.kstep; j.arg[9] = px.nt; j.arg[10] = Kp;
      }
      off += (size_t)3 * d->Cin * Kp;
    }
  return n;
}

static int split_multi_any(const evk_split_job* jobs_dev, const int32_t* block_map_dev, int32_t nblocks,
                           const uint32_t* absmax_dev, void* stream) {
  EVK_REQUIRE(nblocks >= 0 && (nblocks == 0 || (jobs_dev && block_map_dev)), EVK_E_INVALID, "split_multi: bad argument");
  if (nblocks == 0) return EVK_OK;
  hipLaunchKernelGGL(split_weight_multi_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, jobs_dev,
                     block_map_dev, absmax_dev);
  return check_launch("split_weight_multi");
}
extern "C" int evk_conv2d_split_multi(const evk_split_job* jobs_dev, const int32_t* block_map_dev, int32_t nblocks,
                                      void* stream) {
  return split_multi_any(jobs_dev, block_map_dev, nblocks, nullptr, stream);
}
extern "C" int evk_conv2d_split_multi_f16x2(const evk_split_job* jobs_dev, const int32_t* block_map_dev, int32_t nblocks,
                                            const uint32_t* absmax_dev, void* stream) {
  EVK_REQUIRE(absmax_dev, EVK_E_INVALID, "split_multi_f16x2: null scale array");
  return split_multi_any(jobs_dev, block_map_dev, nblocks, absmax_dev, stream);
}

extern "C" int64_t evk_split_job_pairs(const evk_split_job* job) { return job ? (int64_t)split_job_pairs(*job) : 0; }
