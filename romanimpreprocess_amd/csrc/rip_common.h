// Shared declarations of libromanhip (host + device).  gfx950 only.
//
// Floating-point contract of every kernel in this library: one IEEE rounding per written
// operation, in the written order.  The translation units are compiled with -ffp-contract=off
// (hipcc would otherwise fuse a*b+c), IEEE divide/sqrt (hipcc default), no fast-math, f32
// denormals preserved.  Where a fused multiply-add is wanted it is spelled fmaf()/fma().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <map>
#include <vector>

#include "../../include/romanhip.h"

// DQ bits used on the path (SURVEY.md Appendix C / romanimpreprocess_amd/dqflags.py)
#define DQ_DO_NOT_USE 0x1u
#define DQ_SATURATED 0x2u
#define DQ_JUMP_DET 0x4u
#define DQ_NO_FLAT_FIELD (1u << 18)
#define DQ_NO_GAIN_VALUE (1u << 19)
#define DQ_NO_LIN_CORR (1u << 20)
#define DQ_REFERENCE_PIXEL (1u << 31)

#define RIP_CW RIP_CHANNEL_WIDTH

// ---------------------------------------------------------------- device-side plan (ramp fit)
struct RipDiff {
    int32_t i, j;   // difference d[j]-d[i], flag lands on group i (fitting.py:230,249)
    float dt;       // tbar[j]-tbar[i], f32
    float A, B;     // fast-path variance: var ~= A*read^2 + B*dvardt (host f64 sums rounded to f32)
    float inv_dt;   // f32(1/dt), approximate path only
    float relerr;   // bound on the relative error of the approximate significance coming from the variance
    float pad_[1];
};

struct RipVariant {
    int32_t g;      // groups [0,g) take part
    int32_t ndiff, diff_ofs, k_ofs;
    float coef, rfac;
};

struct RipPlanHeader {
    int32_t ngrp, start, nvariants, do_not_flag_first;
    double sa, dsb;        // SthreshA, (SthreshB - SthreshA)
    double loglen;         // np.log(IthreshB / IthreshA), f64
    float ia, ib;          // f32(IthreshA), f32(IthreshB)
    float tbar[RIP_MAX_GROUPS], tau[RIP_MAX_GROUPS], nreads[RIP_MAX_GROUPS];  // nreads as f32(N)
};

// Dense, compile-time-indexable view of the FULL-ramp variant for the register-resident fit
// (fit_full_regs): pair slot ps = 2*(i/2) + (di-1) holds the differences (i, i+di) and (i+1, i+1+di), i even.
struct RipDensePair {
    float inv_dt[2], A[2], B[2];  // per element e: difference (i+e, i+e+di)
    float k1[2];                   // k1 >= 1/(1-r), r = relative error bound of the approximate significance (2 - k1 <= 1/(1+r))
};
struct RipDense {
    uint32_t valid;                    // bit 2*ps + e: that difference is tested (fitting.py:225-229)
    int32_t kidx[2 * RIP_MAX_GROUPS];  // [2*ps + e] index into the compact diff table (exact path)
    float amin;                        // min A over the tested differences (0: some B < 0 -> exact path everywhere)
    float K2[RIP_MAX_GROUPS];          // full-ramp weights
    RipDensePair pairs[RIP_MAX_GROUPS];
};

// layout of the device plan buffer: header | variants[nvariants] | K floats | diffs | dense
struct RipPlan {
    RipPlanHeader h;
    std::vector<RipVariant> variants;
    std::vector<float> kvals;
    std::vector<RipDiff> diffs;
    void *dev = nullptr;          // device copy
    const RipVariant *d_variants = nullptr;
    const float *d_k = nullptr;
    const RipDiff *d_diffs = nullptr;
    const RipDense *d_dense = nullptr;
    RipDense dense;
    size_t bytes = 0;
};

// ---------------------------------------------------------------- device-resident CALDIR of one SCA
struct RipCal {
    bool valid = false;
    int ny = 0, nx = 0, nb = 0;
    int ngrp_dark = 0, ngrp_bias = 0, lin_nplanes = 0;
    int gain_dtype = RIP_F32, ipc_dtype = RIP_F32;
    bool has_amp33 = false, has_ipc = false, has_flat = false, has_bias = false, has_dark_dq = false;
    double refout_slope = 0.0;
    float *dark_data = nullptr;   // (ngrp_dark, ny, nx)
    float *dark_slope = nullptr;  // raw (ny,nx)
    float *dark_rate = nullptr;   // IPC-deconvolved dark rate (ny,nx)  [gen_cal_image.py:217-221]
    uint32_t *dark_dq = nullptr;
    float *read_noise = nullptr;
    float *amp33_med = nullptr;   // (ny,128)
    void *gain = nullptr;         // (ny,nx) f32|f64
    float *lin_coefs = nullptr;   // (nplanes, ny, nx)
    float *lin_smin = nullptr, *lin_smax = nullptr, *lin_sref = nullptr;
    uint32_t *lin_dq = nullptr;
    void *ipc = nullptr;          // (9, ny, nx) embedded in the full frame (border entries unused)
    float *flat_dn = nullptr;     // output of get_flat (ny,nx); border = 1
    uint32_t *flat_flags = nullptr;  // NO_FLAT_FIELD / NO_GAIN_VALUE bits get_flat would OR into pdq
    float *bias = nullptr;        // (ngrp_bias, ny, nx) embedded in the full frame, border = 0
    float *sat_thr = nullptr;     // saturation threshold (ny,nx) or null
    uint32_t *sat_dq = nullptr;   // saturation dq (ny,nx) or null
    // one allocation holding the per-pixel planes the fused kernel walks together, in this order:
    //   [0,NP) Legendre planes | NP Smin | NP+1 Smax | NP+2 Sref | NP+3 lin dq (u32) | NP+4 gain (f32 only)
    //   | NP+5 read noise | NP+6 dark rate | NP+7 flat_dn | NP+8 flat flags (u32)
    //   | NP+9 lin dq OR flat flags | NP+10 lin dq OR dark dq (active region) | NP+11 lin dq OR both   (u32: the flag words the
    //     wave-specialised fused kernel reads in place of NP+3, so that its fit role need not load flat flags / dark dq)
    // (lin_coefs, lin_smin, ..., read_noise, dark_rate, flat_dn, flat_flags point into it)
    float *slab = nullptr;
    // merged_plane[c], c = (flat flags in ? 1 : 0) | (dark dq in ? 2 : 0): plane index relative to NP (3 = the linearity dq alone),
    // or -1 where merging would change the linearity test (an added word carries NO_LIN_CORR / REFERENCE_PIXEL: never seen in a
    // reference-written file; the dispatcher then takes another kernel form)
    int merged_plane[4] = {3, -1, -1, -1};
    size_t bytes = 0;
};

struct rip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;          // reference-pixel pre-pass of the NEXT ramp (device-resident inputs)
    hipEvent_t ev_tab[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    bool ev_done_valid[2] = {false, false};
    int parity = 0;
    bool use_overlap = true;
    int overlap_mode = -1;   // -1: by situation (see rip_calibrate), 1: wherever possible
    bool use_chain2 = true;  // wave-specialised fused kernel where it applies
    int last_form = 0;       // diagnostic: how the last rip_calibrate ran (0 stage kernels, 2 the fused kernel; 1 and 3 were the general and the wave-private fused kernels of rounds 1-2)
    std::string err;
    std::vector<RipCal> cals;
    std::vector<RipPlan *> plans;
    // workspace (grown on demand)
    // per-stage device timing (HIP events on `stream`), see rip_profile_enable / rip_profile_read
    int chain_dbg = 0;
    unsigned long long *chain_dbg_buf = nullptr;  // 4096 waves x 6 phases (diagnostic builds)
    bool use_fused = true;  // rip_set_option("fused", 0) forces the stage-by-stage kernels
    double guard_band = 1e-5;  // relative half-width of the exact-order re-evaluation band of the jump test (rip_set_option_f64)
    bool prof = false;
    std::vector<hipEvent_t> prof_events;  // 6 per rip_calibrate call
    void *ws[18] = {};          // 0-9: calibration path and stage entries; 10-12: Level-1 synthesis (synth.hip); 13: the multi-launch
                                // pre-pass's selection histograms (zero between calls); 14: control words + histograms of the
                                // single-launch pre-pass (zero between calls), 15: its row / channel scratch; 16: the apportioning's list of deferred pixels
    size_t ws_bytes[18] = {};
    void *prepass_stamps = nullptr;   // diagnostic: device buffer of 16 clock stamps per workgroup of the single-launch pre-pass
    bool chain_quad = true;     // a last strip of <= 64 live columns in quad mode (chain2_geometry); false: every strip alike (A/B timing)
    int chain_reserve = 8;      // workgroup slots the 256-column fused kernel leaves free (the next ramp's pre-pass runs in them)
    // reference-pixel tables: -1 = by situation (a pre-pass that overlaps the previous ramp's fused kernel: the nine small launches
    // of refpix.hip, which slip into that kernel's tail; a pre-pass in front of its own ramp on the same stream: the single launch
    // of refpix_one.hip where it covers the frame); 0 = refpix.hip always, 1 = refpix_one.hip wherever it covers the frame
    int prepass_form = -1;
    // rip_calibrate_batch (batch.hip): download stream and the two sets of device buffers, kept between calls
    hipStream_t stream3 = nullptr;
    void *batch_buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t batch_bytes[4] = {0, 0, 0, 0};
    // 1/f frames (pink.hip): transform plan and buffers of the last (length, batch) kept between calls
    void *pink_plan = nullptr, *pink_z = nullptr, *pink_s = nullptr, *pink_tab = nullptr;   // (library plan OR own tables: pink_own)
    bool pink_own = false;
    int pink_form = -1;   // option "pink_form": 0 = the library's transform for every frame length
    hipEvent_t ev_pink = nullptr;   // end of the last 1/f call, on pink_stream: the next call on the OTHER stream waits for it
    hipStream_t pink_stream = nullptr;
    bool ev_pink_valid = false;
    size_t pink_L = 0;
    int pink_chunk = 0;
    // set by the entry points that queue work on `stream` with device pointers (rip_synth_*, rip_stats_*): the next overlapped
    // rip_calibrate then orders its second-stream pre-pass behind that work (it may have produced the call's inputs)
    bool stream_dirty = false;
    // the pre-pass / saturation pass share workspaces (selection histograms, row tables in the making, exceed bits): when two
    // consecutive calls run them on different streams (a non-overlapped call between overlapped ones), the later waits for the
    // earlier through this event
    // launch geometry of the fused kernel, per CONTEXT (a second context may sit on another device): CU count
    int ncu = 0;
    // 1/f frames made AHEAD on the second stream (rip_synth_frames_ahead, pink.hip) for the next rip_synth_fill: the transforms
    // (HBM-bound) then run beside the apportioning and the inverse-linearity kernels (arithmetic-bound) of the same exposure
    hipEvent_t ev_frames = nullptr, ev_fill = nullptr, ev_ahead = nullptr;
    bool frames_pending = false, ev_fill_valid = false;
    uint64_t frames_seed = 0;
    int frames_geom[3] = {0, 0, 0};   // rows, channel width, frames
    std::vector<double> share_tab;   // synth.hip: the read-share table whose device copy sits in workspace slot 10
    hipEvent_t ev_pre = nullptr;
    hipStream_t pre_stream = nullptr;
    bool ev_pre_valid = false;
    hipEvent_t ev_in = nullptr;
    int batch_completed = 0;  // of the last rip_calibrate_batch: ramps completed (all of them unless it returned an error)
};

// ---------------------------------------------------------------- host helpers
int rip_fail(rip_ctx *ctx, int code, const char *fmt, ...);
void rip_pink_release(rip_ctx *ctx);   // pink.hip: drops the cached transform plan and buffers
void *rip_ws(rip_ctx *ctx, int slot, size_t bytes);  // nullptr on failure (error recorded)

#define RIP_HIP(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) return rip_fail(ctx, RIP_EHIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------- kernel launchers (host side)
// rampfit.hip
struct RampFitArgs {
    const float *cube;      // (G, ny, nx) corrected cube
    const uint8_t *gdq_in;  // (G, ny, nx)
    uint8_t *gdq_out;       // (G, ny, nx) or nullptr
    const uint32_t *pdq_in; // (ny, nx)
    uint32_t *pdq_out;      // (ny, nx)
    const void *gain;       // (ny, nx)
    const float *read_noise;
    float *slope, *err_read, *err_poisson;
    // finish (any may be null -> step skipped)
    int finish;                  // 0: raw ramp_fit outputs (fitting.ramp_fit); 1: A11 algebra (+dark, +flat)
    const float *dark_rate;      // subtract on the active region
    const uint32_t *dark_dq;
    const float *flat;           // divide (already flat_dn / area as f32)
    const uint32_t *flat_flags;  // OR into pdq
    int ny, nx, nb, ngrp;
};
int rip_launch_rampfit(rip_ctx *ctx, const RipPlan *plan, const RampFitArgs &a, int gain_dtype);
// one jump_detect pass over the plan's full ramp; rdq updated in place, smap (ndiff, ny, nx) out (device pointers)
int rip_launch_jumpdetect(rip_ctx *ctx, const RipPlan *plan, const float *cube, uint8_t *rdq, const void *gain, int gain_dtype,
                          const float *read_noise, float *slope, float *err_read, float *err_poisson, float *smap, int ny,
                          int nx, int nb);

// linearity.hip
struct LinArgs {
    const void *data;        // (G, ny, nx) u16 or f32
    int data_dtype;
    float *phi;              // (G, ny, nx)
    const uint8_t *gdq;      // for attempt_corr = group not saturated; nullptr -> attempt everything
    int gdq_is_attempt;      // 1: gdq holds attempt_corr bytes (nonzero = attempt) instead of groupdq
    const uint32_t *pdq_in;  // may be null (treated as 0)
    uint32_t *pdq_out;       // pdq_in | dq_lin
    // refpix apply (null rowcorr -> skipped)
    const float *dark_data;  // (>=G, ny, nx)
    const double *rowcorr;   // (G, ny)   slope * f64(f32(ref_med - ctr))
    const double *lines;     // (G, nx/128, 2)  (m, c)
    // bias (null -> skipped); embedded full-frame planes, group offset applied by the caller
    const float *bias;
    // linearity (null coefs -> skipped: phi = data after refpix/bias)
    const float *coefs, *smin, *smax, *sref;
    const uint32_t *lin_dq;
    int nplanes;
    int do_not_flag_first;
    int ny, nx, nb, ngrp;
};
int rip_launch_lin(rip_ctx *ctx, const LinArgs &a);

// chain.hip (the fused kernel)
struct ChainArgs {
    const void *data;  // (G, ny, nx) u16 | f32
    int data_u16;
    const uint8_t *gdq;   // (G, ny, nx)
    const uint32_t *pdq;  // (ny, nx)
    // reference-pixel tables (rowcorr null -> step skipped)
    const float *dark_data;
    const double *rowcorr, *lines;
    const double *rowcorr_t;  // (ny, G) copy of rowcorr (wave-private kernel: one wide scalar load per row)
    const float *bias;  // embedded planes, already offset to the first group used; null -> skipped
    const float *planes;  // RipCal::slab
    int do_not_flag_first;
    const void *kern;   // (9, ny, nx) embedded ipc4d
    int finish;
    int dark_rate;        // 1: subtract plane NP+6 on the active region
    const uint32_t *dark_dq;
    const float *flat;    // plane NP+7 or the per-exposure f32(flat_dn/area) plane; null -> no flat step
    float *slope, *err_read, *err_poisson;
    uint32_t *pdq_out;
    uint8_t *gdq_out;  // may be null
    float *cube_out;   // may be null
    const RipDense *dense;        // RipPlan::d_dense
    unsigned long long *dbg_buf;  // CH_STAMP builds only: per-wave phase cycle sums
    int merged_dq;     // plane index (relative to NP) of the flag word that already holds flat flags / dark dq as this call applies them
    int dbg;           // timing experiments only (rip_set_option "chain_dbg"): skips phases, results invalid
    int ny, nx, nb, ngrp;
    // launch geometry (set by the launcher, chain2_kernel.h): geo_nr row ranges of geo_rows rows for each full-width strip; where
    // the last strip has at most 64 live columns (nx = 4096: 17th strip of the 256-column form) it is covered by geo_nq workgroups
    // whose four wave columns each march down their OWN range of geo_rows_q rows of that strip (0: every strip alike)
    int geo_nr, geo_rows, geo_nq, geo_rows_q;
};
bool rip_chain_supported(const rip_ctx *ctx, int nplanes, int G, int k_dtype, int gain_dtype);
int rip_launch_chain(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a, int nplanes, int k_dtype);

// ipc.hip
struct IpcArgs {
    const float *in;   // (G, ny, nx)
    float *out;        // (G, ny, nx)   (border copied through)
    const void *kern;  // (9, ny, nx) embedded
    const void *gain;  // (ny, nx) or nullptr (= 1)
    int k_dtype, g_dtype;
    int ny, nx, nb, ngrp;
};
int rip_launch_ipc_cube(rip_ctx *ctx, const IpcArgs &a);
// generic single-image forward / reverse operator (stage API + CALDIR-derived planes)
int rip_launch_ipc_image(rip_ctx *ctx, int reverse, int order, const void *img, int img_dtype, int ny, int nx,
                         const void *kern /* (9,ny,nx) */, int k_dtype, const void *gain, int g_dtype, void *out,
                         int gain_div_clip /*unused*/);

// refpix.hip
struct RefpixArgs {
    const void *data;  // (G, ny, nx) u16|f32
    int data_dtype;
    const float *dark_data;
    const uint16_t *amp33;   // (G, ny, 128)
    const float *amp33_med;  // (ny, 128)
    double slope;
    const double *lines_override;  // device (G, nch, 2) or nullptr
    double *rowcorr;  // out (G, ny)
    double *rowcorr_t;  // out (ny, G): the same values, row-major (may be null)
    double *lines;    // out (G, nch, 2)
    int ny, nx, ngrp;
    int background = 0;   // 1: launched beside the previous ramp's fused kernel (second stream)
    hipStream_t stream = nullptr;   // where the launches go (null: the context's main stream)
};
int rip_launch_refpix_prepass(rip_ctx *ctx, const RefpixArgs &a);
// refpix_one.hip: the same tables in one launch (frames up to 4096 rows with a reference output)
bool rip_refpix_one_supported(const RefpixArgs &a);
int rip_launch_refpix_one(rip_ctx *ctx, const RefpixArgs &a);
int rip_refpix_one_status(rip_ctx *ctx, int *status);
// the general forms (any argument of reference_subtraction.py's two functions); device pointers
int rip_refpix_row_general(rip_ctx *ctx, float *d_image, int ny, int width, int nside, int use_ref_channel, int mode,
                           double slope, float *d_ref_med, float *d_sci_med, float *d_ctr);
int rip_refpix_channel_general(rip_ctx *ctx, float *d_image, int ny, int width, int channel_start, int channel_end, int nchan,
                               const double *d_lines, float *d_bottom_top);
int rip_refpix_image(rip_ctx *ctx, float *d_image, int ny, int nx, double slope, int do_row, int do_channel,
                     const double *d_lines, float *d_ref_med, float *d_ctr, float *d_bottom_top);

// misc.hip
int rip_launch_embed(rip_ctx *ctx, const void *src, void *dst, int nplanes, int ny, int nx, int nb, int elem_size);
int rip_launch_flat_prepare(rip_ctx *ctx, const float *flat, const void *gain, int g_dtype, int ny, int nx, int nb,
                            float *flat_padded, void *gain_clipped, uint32_t *flags, int with_gain);
int rip_launch_flat_area(rip_ctx *ctx, const float *flat_dn, const double *area, float *out, size_t n);
// out = lin_dq | (flat_flags or 0) | (dark_dq on the active region or 0); *d_clash |= added bits & (NO_LIN_CORR | REFERENCE_PIXEL)
int rip_launch_merge_dq(rip_ctx *ctx, const uint32_t *lin_dq, const uint32_t *flat_flags, const uint32_t *dark_dq, uint32_t *out, int ny,
                        int nx, int nb, uint32_t *d_clash);
int rip_launch_or_bytes(rip_ctx *ctx, uint8_t *bytes, size_t n, uint8_t bit, hipStream_t stream = nullptr);
// dq-init + saturation flagging (misc.hip): gdq_in / pdq_in may be null (= zeros)
int rip_launch_satflag(rip_ctx *ctx, const void *data, int data_dtype, const float *thr, const uint32_t *sat_dq,
                       const uint8_t *gdq_in, const uint32_t *pdq_in, uint8_t *gdq_out, uint32_t *pdq_out, int G, int ny,
                       int nx, int backup, int skip_firstn, int dnu_first, const double *dilution = nullptr,
                       hipStream_t stream = nullptr);
