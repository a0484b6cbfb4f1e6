/*
 * romanhip.h -- C-ABI of libromanhip.so: the MI355X (gfx950) implementation of the per-pixel
 * L1->L2 detector-calibration path of romanimpreprocess.
 *
 * The reference has no FFI layer; its seam is the Python-function level of
 *   src/romanimpreprocess/utils/reference_subtraction.py:16,77   (ref_subtraction_channel/_row)
 *   src/romanimpreprocess/utils/ipc_linearity.py:37,102,145,276  (ipc_fwd, ipc_rev, correct_cube, multilin)
 *   src/romanimpreprocess/utils/fitting.py:89,258                (jump_detect, ramp_fit)
 *   src/romanimpreprocess/utils/flatutils.py:20                  (get_flat)
 *   src/romanimpreprocess/L1_to_L2/gen_cal_image.py:531-629      (the in-line per-pixel arithmetic of calibrateimage)
 * Each entry point below names the reference callable it replaces.  The Python binding a
 * reference maintainer would add is shown in INTEGRATION.md (ctypes).
 *
 * Conventions
 *   - plain C types only; all arrays are C-order, dense.
 *   - a frame has `ny` rows and `nx` science columns (4096 x 4096 in flight); the reference
 *     pixel border is `nborder` (4); the active region is (ny-2nb) x (nx-2nb); the reference
 *     output ("amp33") has `ny` rows x 128 columns.  nx must be a multiple of 128 when the
 *     reference-pixel stage is used.
 *   - the caller owns every buffer it passes; the library copies CALDIR arrays to the device at
 *     rip_caldir_upload and owns those copies.
 *   - `location` says where ramp/output buffers live: RIP_HOST (the library stages them through
 *     HBM) or RIP_DEVICE (pointers into this process's HIP address space, e.g. torch tensors).
 *   - every function returns 0 on success, a negative rip_status otherwise; rip_last_error()
 *     gives the message.  No exceptions or longjmp cross this boundary.
 *   - one rip_ctx per GPU; a ctx is not thread-safe (one calling thread at a time).
 */
#ifndef ROMANHIP_H
#define ROMANHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RIP_VERSION 100 /* 0.1.0 */
/* rip_version() of a library compiled with -DRIP_TIMING_BUILD (timing experiments whose kernels may skip phases: results
   invalid by construction) = RIP_VERSION + RIP_TIMING_BUILD_FLAG; the Python binding refuses such a library unless
   ROMANHIP_ALLOW_TIMING_BUILD=1 */
#define RIP_TIMING_BUILD_FLAG 1000000
#define RIP_MAX_GROUPS 64
#define RIP_CHANNEL_WIDTH 128

typedef struct rip_ctx rip_ctx;

typedef enum { RIP_OK = 0, RIP_EINVAL = -1, RIP_ENOMEM = -2, RIP_EHIP = -3, RIP_ESTATE = -4 } rip_status;
typedef enum { RIP_F32 = 0, RIP_F64 = 1, RIP_U16 = 2 } rip_dtype;
typedef enum { RIP_HOST = 0, RIP_DEVICE = 1 } rip_location;
/* rip_ramp_desc::inputs_ready: what the library may assume about DEVICE-resident inputs when rip_calibrate is called */
typedef enum { RIP_INPUTS_STREAM_ORDERED = 0, RIP_INPUTS_COMPLETE = 1 } rip_inputs_ready;

/* stages of the chain (bit mask), in the order of calibrateimage (gen_cal_image.py:531-629) */
enum {
    RIP_STAGE_REFPIX = 1 << 0,  /* :531-556  reference-pixel row/channel correction              */
    RIP_STAGE_BIAS = 1 << 1,    /* :559-565  data[act] -= biascorr                                  */
    RIP_STAGE_LIN = 1 << 2,     /* :580-588  multilin + pdq |= dq_lin                               */
    RIP_STAGE_IPC = 1 << 3,     /* :594-597  correct_cube                                           */
    RIP_STAGE_RAMPFIT = 1 << 4, /* :600      ramp_fit (slope, errors, jump flags, flag propagation) */
    RIP_STAGE_DARK = 1 << 5,    /* :603      slope[act] -= IPC-deconvolved dark rate                 */
    RIP_STAGE_FLAT = 1 << 6,    /* :616-629  get_flat flags + divide by flat/AreaFactor              */
    RIP_STAGE_ALL = 0x7f
};

/* ---- CALDIR arrays of one SCA (host pointers) ------------------------- */
typedef struct {
    int32_t ny, nx, nborder;
    /* dark file: data (ngrp_dark,ny,nx) f32, dark_slope (ny,nx) f32, dq (ny,nx) u32 or NULL */
    int32_t ngrp_dark;
    const float *dark_data;
    const float *dark_slope;
    const uint32_t *dark_dq;
    /* read file: data (ny,nx) f32; amp33.med (ny,128) f32 or NULL; refout_slope = the scalar of
       gen_cal_image.py:542-553 (computed by the host from M_PINK, RU_PINK, C_PINK, median(std)) */
    const float *read_noise;
    const float *amp33_med;
    double refout_slope;
    /* gain file: data (ny,nx), f32 or f64 */
    const void *gain;
    int32_t gain_dtype;
    /* linearitylegendre file: data (lin_nplanes,ny,nx) f32; Smin,Smax,Sref (ny,nx) f32; dq u32 */
    int32_t lin_nplanes;
    const float *lin_coefs;
    const float *lin_smin, *lin_smax, *lin_sref;
    const uint32_t *lin_dq;
    /* ipc4d file: data (3,3,ny-2nb,nx-2nb), f32 or f64; NULL = no IPC correction */
    const void *ipc4d;
    int32_t ipc_dtype;
    /* flat (pflat) file: data (ny,nx) f32; NULL = no flat division */
    const float *flat;
    /* biascorr file: data (ngrp_bias,ny-2nb,nx-2nb) f32; NULL = skip */
    int32_t ngrp_bias;
    const float *biascorr;
    /* saturation file: data (ny,nx) f32 threshold in DN, dq (ny,nx) u32 (NO_SAT_CHECK honoured) or NULL.
       Only needed for rip_ramp_desc::flag_saturation; NULL = not available. */
    const float *saturation;
    const uint32_t *saturation_dq;
} rip_caldir_desc;

/* ---- ramp-fit plan: MA table, weights, thresholds (host scalars) --- */
typedef struct {
    int32_t ngrp;
    int32_t exclude_first;     /* config EXCLUDE_FIRST (gen_cal_image.py:142, fitting.py:159-161)     */
    int32_t do_not_flag_first; /* read_pattern[0] == [0] (gen_cal_image.py:583-584)                   */
    float tbar[RIP_MAX_GROUPS]; /* meta["tbar"], f32 (gen_cal_image.py:137)                           */
    float tau[RIP_MAX_GROUPS];  /* meta["tau"],  f32 (:138)                                            */
    int16_t nreads[RIP_MAX_GROUPS]; /* meta["N"], int16 (:135)                                          */
    float K[RIP_MAX_GROUPS];    /* fitting.construct_weights(...) cast f32 (fitting.py:86)             */
    /* per fit variant v: v = 0 is the full ramp; v = 1.. are ramps truncated to [0, g_v) for
       g_v = ngrp-1 ... 3+start (fitting.py:326).  coef = Poisson coefficient (fitting.py:196-200),
       rfac = sqrt(sum K^2/N) (fitting.py:209); both f32 scalars computed by the host exactly as the
       reference computes them. */
    int32_t nvariants;
    int32_t variant_g[RIP_MAX_GROUPS];
    float variant_coef[RIP_MAX_GROUPS];
    float variant_rfac[RIP_MAX_GROUPS];
    /* jump thresholds (fitting.py:172-184) */
    double sthresh_a, sthresh_b, ithresh_a, ithresh_b;
} rip_plan_desc;

/* ---- one ramp in, one calibrated image out ---------------------------------------------- */
typedef struct {
    int32_t location;      /* rip_location of every pointer in this struct                             */
    int32_t ngrp;
    const void *data;      /* (ngrp,ny,nx) u16 (Level-1) or f32 (after dq-init)                        */
    int32_t data_dtype;    /* RIP_U16 or RIP_F32                                                       */
    const uint16_t *amp33; /* (ngrp,ny,128) or NULL                                                    */
    const uint8_t *groupdq;  /* (ngrp,ny,nx) after dq-init + saturation flagging                       */
    const uint32_t *pixeldq; /* (ny,nx)                                                                */
    const double *area_factor; /* (ny,nx) f64 pixel-area ratio (gen_cal_image.py:618-622) or NULL = 1  */
    /* optional override of the channel-step line fit: (ngrp,nx/128,2) f64 (m,c) computed by the
       host with LAPACK exactly as reference_subtraction.py:57-60; NULL = fitted on the device */
    const double *channel_lines;
    /* dq-init + saturation flagging on the device before the chain (gen_cal_image.py:148-185: romancal's dq_init and
       saturation steps with n_pix_grow_sat = 1; stcal's source is not in the reference tree, so this follows the
       documented behaviour restated in romanimpreprocess_amd/L1_to_L2/gen_cal_image.py:flag_saturation -- PARITY
       UNPINNED).  flag_saturation != 0: group g >= sat_skip_firstn is SATURATED where data >= threshold in the 3x3
       neighbourhood, sticky for later groups and set on the sat_backup preceding groups; pixeldq |= SATURATED where any
       group is flagged; DO_NOT_USE is set on group 0 when the plan excludes the first group.  groupdq may then be NULL
       (= zeros); pixeldq is the mask dq. */
    int32_t flag_saturation;
    int32_t sat_backup;      /* config SATURATION_BACKUP (gen_cal_image.py:172), default 1 */
    int32_t sat_skip_firstn; /* 1 in the reference's call */
    /* groups that average several reads: a pixel whose LAST reads saturate holds a group value below the threshold, so
       the group is compared with threshold * sat_dilution[g], sat_dilution[g] = mean(read_pattern[g]) / read_pattern[g][-1]
       (stcal's read_pattern rule, "checks for groups with some reads saturated", docs/L1_to_L2_README.rst:139-141; the
       reference hands the read pattern over at gen_cal_image.py:172-185).  Host pointer to ngrp doubles (also for device
       ramps), or NULL = 1 for every group.  A NaN entry (0/0 for a group holding only read 0) never flags. */
    const double *sat_dilution;
    /* When the DEVICE-resident inputs of this call (data, amp33, groupdq, pixeldq, area_factor, channel_lines) are valid.
       The reference-pixel pre-pass of a call reads data / amp33 on a SECOND stream so that it can run beside the previous
       call's main kernel; these two fields say what that stream has to wait for.  Ignored for location == RIP_HOST.
         inputs_ready == RIP_INPUTS_STREAM_ORDERED (0, the default): the inputs are complete, or are written by work the
             caller has ALREADY queued on rip_stream().  Every kernel of the call, the pre-pass included, is ordered behind
             everything queued on rip_stream() at the time of the call (so the pre-pass cannot overlap the previous call's
             main kernel: correct for every caller that uses one stream, ~0.03-0.06 ms per 4096^2 x 8 ramp slower).
         inputs_ready == RIP_INPUTS_COMPLETE: the caller guarantees that the inputs are complete NOW (a ramp resident since
             the last synchronisation, ...): the pre-pass starts at once.
         ready_event != NULL (a hipEvent_t as void*, recorded by the caller behind the work that writes the inputs, on any
             stream): the pre-pass and the main kernels wait for exactly that event; inputs_ready is not looked at.  This is
             the form for inputs produced on another stream (a torch side stream, a copy engine) without giving up the
             overlap.  The event must stay alive until the call returns (the wait is queued before that). */
    int32_t inputs_ready;
    void *ready_event;
    /* location == RIP_HOST only: OR DO_NOT_USE into groupdq[0] on the library's device copy (gen_cal_image.py:142-143 does it to
       the array itself when the first group is excluded; a host caller that wants its own array untouched would otherwise copy
       134 MB to set one plane's bit).  The groupdq returned carries the bit, as the reference's does. */
    int32_t or_first_group;
} rip_ramp_desc;

typedef struct {
    int32_t location;
    float *slope;        /* (ny,nx) DN/s                                  */
    float *err_read;     /* (ny,nx)                                       */
    float *err_poisson;  /* (ny,nx)                                       */
    uint32_t *pixeldq;   /* (ny,nx)                                       */
    uint8_t *groupdq;    /* (ngrp,ny,nx) or NULL                          */
    float *cube;         /* (ngrp,ny,nx) corrected cube (after the last cube stage run) or NULL */
} rip_outputs;

/* ---- life cycle ------------------------------------------------------------------------- */
int rip_version(void);
int rip_ctx_create(int device_id, rip_ctx **out);
void rip_ctx_destroy(rip_ctx *ctx);
const char *rip_last_error(const rip_ctx *ctx); /* ctx may be NULL: message of the last failed create */
int rip_synchronize(rip_ctx *ctx);
/* the HIP stream all work of this ctx is enqueued on (hipStream_t as void*) */
void *rip_stream(rip_ctx *ctx);
/* page-locked host memory for arrays handed to the library with location RIP_HOST (the reference's numpy arrays are
   pageable; buffers from here are copied at PCIe rate).  NULL on failure (rip_last_error). */
void *rip_host_alloc(rip_ctx *ctx, size_t bytes);
void rip_host_free(rip_ctx *ctx, void *p);

/* CALDIR: replaces the per-call asdf.open(caldir[...]) of the reference with device-resident
   copies (plus the per-SCA constants derived from them: IPC-deconvolved dark rate
   gen_cal_image.py:217-221 and the flat of flatutils.get_flat) */
int rip_caldir_upload(rip_ctx *ctx, int sca_slot, const rip_caldir_desc *desc);
int rip_caldir_drop(rip_ctx *ctx, int sca_slot);

/* plan: replaces meta{ngrp,N,tbar,tau,K,jump_detect_pars} of gen_cal_image.py:123-145,439-444 */
int rip_plan_create(rip_ctx *ctx, const rip_plan_desc *desc, int *plan_id);
int rip_plan_destroy(rip_ctx *ctx, int plan_id);

/* the chain: replaces gen_cal_image.py:531-629 (stages selects a sub-chain).  Asynchronous with
   respect to the host when location == RIP_DEVICE (use rip_synchronize / the stream).
   Device-resident INPUTS: by default the call is ordered behind everything already queued on rip_stream() (inputs written
   by a kernel the caller queued there are safe without a synchronisation); rip_ramp_desc::inputs_ready / ready_event let a
   caller whose inputs are complete, or guarded by an event of another stream, keep the overlap of the reference-pixel
   pre-pass (second stream) with the previous call's main kernel.  Outputs are ordered on rip_stream() either way; with
   "overlap" off every kernel of a call runs on rip_stream(). */
int rip_calibrate(rip_ctx *ctx, int sca_slot, int plan_id, unsigned stages, const rip_ramp_desc *in,
                  const rip_outputs *out);

/* A batch of n ramps in HOST memory through the same chain, pipelined over PCIe: upload of ramp i+1, chain of ramp i and
   download of ramp i-1 overlap (three streams, two sets of device buffers); results equal those of n single calls.  What a
   batch driver (runs/summer2025run/OpenUniverse_to_L1L2.py:123-137: every SCA of an exposure) hands its arrays to.  All
   ramps share the CALDIR slot, the plan, the group count and the data dtype; location must be RIP_HOST in every
   descriptor; the stage mask must include the ramp fit; outputs::cube is not available here.  Page-locked arrays
   (rip_host_alloc) make the copies run at PCIe rate in both directions at once.  Returns when everything has arrived. */
int rip_calibrate_batch(rip_ctx *ctx, int sca_slot, int plan_id, unsigned stages, int n, const rip_ramp_desc *in,
                        const rip_outputs *out);
/* after rip_calibrate_batch: the number of ramps (from the front of the batch) whose outputs are complete.  n on success;
   after an error return every stream has been drained, outputs [0, this) are valid and the rest are not. */
int rip_calibrate_batch_completed(rip_ctx *ctx);

/* ---- stage-level entry points (host arrays; for function-level drop-in and parity tests) ----- */

/* reference_subtraction.ref_subtraction_row(image, use_ref_channel=True, slope) followed by
   ref_subtraction_channel(image, use_ref_channel=True) on one (ny, nx+128) f32 image, in place.
   do_row / do_channel select the steps; medians out (optional): ref_med (ny) f32, ctr (1) f32,
   bottom_top (nx/128+1, 2) f32; lines in (optional, (nx/128+1,2) f64 (m,c)) override the fit. */
int rip_stage_refpix_image(rip_ctx *ctx, float *image, int ny, int nx, double slope, int do_row, int do_channel,
                           const double *lines, float *ref_med, float *ctr, float *bottom_top);

/* reference_subtraction.ref_subtraction_row with ANY of its arguments (reference_subtraction.py:77-125) on one (ny, width) f32
   image, in place.  nside = science columns (the reference hard-codes 4096): border pixels 0:4 and nside-4:nside, science
   pixels 4:nside-4, reference output nside:nside+128.  use_ref_channel: the row medians are those of the reference output
   (width >= nside+128), else of the 4+4 border pixels (:108-111).  mode:
     RIP_ROW_MEDIANS_ONLY  no update: ref_med / sci_med / ctr out -- what the host needs for the np.polyfit of slope=None (:114)
     RIP_ROW_SLOPE_F64     image = f32(f64(image) - slope * f64(f32(ref_med - ctr)))   (slope a numpy f64 scalar, :123)
     RIP_ROW_SLOPE_F32     image = image - f32(slope) * (ref_med - ctr), every operation f32 (slope a Python float or a numpy
                           f32 scalar, as np.polyfit returns it for f32 medians: numpy's promotion rules)
   ref_med, sci_med (ny) f32 and ctr (1) f32 out, each may be NULL (sci_med costs a 4088-value selection per row). */
enum { RIP_ROW_MEDIANS_ONLY = 0, RIP_ROW_SLOPE_F64 = 1, RIP_ROW_SLOPE_F32 = 2 };
int rip_stage_refpix_row(rip_ctx *ctx, float *image, int ny, int width, int nside, int use_ref_channel, int mode, double slope,
                         float *ref_med, float *sci_med, float *ctr);

/* reference_subtraction.ref_subtraction_channel with ANY of its arguments (:16-74): nchan windows of columns
   [channel_start + 128 k, channel_end + 128 k), k = 0 .. nchan-1 (32, or 33 with use_ref_channel), handled one after the
   other as the reference's loop does (windows wider than 128 columns overlap and see each other's updates); per window
   the medians of rows 0:4 and ny-4:ny, the line through (1.5, b), (ny-2.5, t) -- or lines (nchan,2) f64 (m,c) from the
   caller's LAPACK -- subtracted from every row.  bottom_top (nchan,2) f32 out or NULL. */
int rip_stage_refpix_channel(rip_ctx *ctx, float *image, int ny, int width, int channel_start, int channel_end, int nchan,
                             const double *lines, float *bottom_top);

/* The reference-pixel TABLES the chain applies to a ramp (gen_cal_image.py:531-556 through reference_subtraction.py:16-125,
   with the reference output: use_ref_channel=True, slope given), from host arrays: data (ngrp,ny,nx) u16|f32, dark (>= ngrp,ny,nx)
   f32, amp33 (ngrp,ny,128) u16, amp33_med (ny,128) f32 -> rowcorr (ngrp,ny) f64 = slope * f64(f32(row median - ctr)) and
   lines (ngrp,nx/128,2) f64 = (m, c) of the science channels.  form: 1 = the single-launch kernel (refpix_one.hip; frames up to
   4096 rows), 0 = the multi-launch kernels (refpix.hip), -1 = what rip_calibrate takes for a pre-pass in front of its own ramp
   (option "prepass_form").  Both forms give identical bits.  status out (may be NULL): != 0 when a group barrier of the single-launch kernel timed out. */
int rip_stage_refpix_tables(rip_ctx *ctx, const void *data, int data_dtype, const float *dark, const uint16_t *amp33,
                            const float *amp33_med, double slope, int ngrp, int ny, int nx, int form, double *rowcorr,
                            double *lines, int *status);

/* ipc_linearity.multilin: S (ngrp,ny,nx) f32 -> phi (ngrp,ny,nx) f32, dq (ny,nx) u32.
   attempt_corr (ngrp,ny,nx) u8 nonzero = flag when extrapolated, or NULL = all. */
int rip_stage_multilin(rip_ctx *ctx, const float *S, int ngrp, int ny, int nx, int nplanes, const float *coefs,
                       const float *smin, const float *smax, const float *sref, const uint32_t *lin_dq,
                       int do_not_flag_first, const uint8_t *attempt_corr, float *phi, uint32_t *dq);

/* ipc_linearity.ipc_fwd / ipc_rev on one (ny,nx) image with kernel (3,3,ny,nx); gain NULL or (ny,nx).
   image/out dtype = img_dtype (f32|f64); kernel k_dtype; gain g_dtype; out dtype = promote(all). */
int rip_stage_ipc_image(rip_ctx *ctx, int reverse, int order, const void *image, int img_dtype, int ny, int nx,
                        const void *kernel, int k_dtype, const void *gain, int g_dtype, void *out);

/* ipc_linearity.correct_cube: data (ngrp,ny,nx) f32 in place; kernel (3,3,ny-2nb,nx-2nb); gain (ny,nx) or NULL */
int rip_stage_correct_cube(rip_ctx *ctx, float *data, int ngrp, int ny, int nx, int nb, const void *kernel,
                           int k_dtype, const void *gain, int g_dtype);

/* fitting.ramp_fit: data (ngrp,ny,nx) f32; rdq u8 and pdq u32 updated in place; outputs (ny,nx) f32 */
int rip_stage_ramp_fit(rip_ctx *ctx, int plan_id, const float *data, uint8_t *rdq, uint32_t *pdq, int ny, int nx,
                       int nb, const void *gain, int g_dtype, const float *read_noise, float *slope,
                       float *err_read, float *err_poisson);

/* fitting.jump_detect (fitting.py:89-255): ONE pass of slope fit + jump flagging over all ngrp groups of the plan with the
   plan's weights K.  data (ngrp,ny,nx) f32; rdq (ngrp,ny,nx) u8 updated in place (JUMP_DET OR-ed into group i of a flagged
   difference, active region only: the reference ORs into whatever flag cube it is handed); slope / err_read / err_poisson
   (ny,nx) f32 out; smap (2*(ngrp-start)-3, ny, nx) f32 out: the significance of every tested difference, every pixel, in
   the reference's exact operation order.  truncate_ramp = t of the reference is this call under a plan of t groups holding
   the two-point weights of fitting.py:162-167 (the Python mirror builds it). */
int rip_stage_jump_detect(rip_ctx *ctx, int plan_id, const float *data, uint8_t *rdq, int ny, int nx, int nb, const void *gain,
                          int g_dtype, const float *read_noise, float *slope, float *err_read, float *err_poisson, float *smap);

/* flatutils.get_flat: flat (ny,nx) f32, gain, kernel; pdq updated in place (may be NULL) */
int rip_stage_get_flat(rip_ctx *ctx, const float *flat, int ny, int nx, int nb, const void *gain, int g_dtype,
                       const void *kernel, int k_dtype, int ipc_deconvolve, uint32_t *pdq, float *out);

/* ---- post-path 2-D reductions: between the chain's outputs and the L2 file -------------------
   Array arguments of this section, of the noise-layer section (rip_stage_noise_inject, rip_stage_poisson_resample) and of
   rip_stage_pearson may be host arrays OR device pointers (each is copied with hipMemcpyDefault): a driver that keeps its
   planes in HBM (the noise-layer loop, romanimpreprocess_amd/L1_to_L2/gen_noise_image.py) hands them over without a PCIe
   round trip.  The calls stay synchronous: they return when the result is complete.  Small tables (ranks, counts, Legendre
   tables, coefficients, weights) are host arrays. */

/* maskhandling.CombinedMask.build (maskhandling.py:82-117): grow[bit] in {0, 1, 5, 9, 25} = how the layer of that dq bit
   is grown (copy, plus, 3x3, 5x5; zero padded); mask (ny,nx) u8 = 1 where masked.  Exact. */
int rip_stage_build_mask(rip_ctx *ctx, const uint32_t *dq, int ny, int nx, const uint8_t grow[32], uint8_t *mask);

/* SLICEOUT endslice (gen_cal_image.py:697-712): (ny-2nb, nx-2nb) i8, iend-1 of the last group that first saturates, -1 if
   none.  Exact. */
int rip_stage_endslice(rip_ctx *ctx, const uint8_t *rdq, int ngrp, int ny, int nx, int nb, int8_t *out);

/* sky.binkxk(np.where(mask, nan, arr), k) (sky.py:20-41): (ny/k, nx/k) f32 block means, NaN where the block holds a
   masked or NaN pixel; mask may be NULL.  f32 sums in row-then-column order (numpy's order differs: ~1e-7 relative). */
int rip_stage_bin_mean(rip_ctx *ctx, const float *arr, const uint8_t *mask, int ny, int nx, int k, float *out);

/* Order statistics ignoring NaN, per block of ky x kx pixels (nby x nbx blocks from (y0,x0)) of an (ny,nx) f32 image:
   the building block of np.nanpercentile (sky.py:72-74) and of the block nan-medians of medfit (sky.py:152).
   counts[blk] = non-NaN elements; vals[blk*nranks + r] = element of 0-based ascending rank ranks[blk*nranks + r] (NaN
   when out of range).  ranks may be NULL with nranks = 0 (counts only).  Exact. */
int rip_stage_select_ranks(rip_ctx *ctx, const float *arr, int ny, int nx, int y0, int x0, int ky, int kx, int nby, int nbx,
                           int nranks, const int64_t *ranks, int64_t *counts, float *vals);

/* smoothed histogram of smooth_mode (sky.py:80-84): out[i] = sum over non-NaN x of exp(-0.5 ((z[i]-x)/scale)^2), nz <= 32,
   f64 (summation order differs from numpy: ~1e-13 relative). */
int rip_stage_gauss_hist(rip_ctx *ctx, const float *arr, int64_t n, const double *z, int nz, double scale, double *out);

/* Legendre model of medfit (sky.py:183-191): model = f32(sum_k coef[k] * outer(LPY[j_k], LPX[i_k])) accumulated in f64 in
   the order k = 0.. with (i, j): i = 0..order, j = 0..order-i; LPX (order+1, nx), LPY (order+1, ny) f64 from the host.
   subtract != 0: arr -= model in place; model_out (ny,nx) f32 optional.  Exact. */
int rip_stage_legendre2d(rip_ctx *ctx, float *arr, int ny, int nx, int order, const double *LPX, const double *LPY,
                         const double *coef, int subtract, float *model_out);

/* ---- simulation side ------------------------------------------------- */
/* ipc_linearity.invlinearity (ipc_linearity.py:347-394): 24 bisection steps on z in (-1, 1) of the Legendre series evaluated
   as ipc_linearity._lin does without the extrapolation branch, then S = Smin + (Smax - Smin)/2 * (1 + z).  slin (ny,nx) f32 or
   f64 (dtype); coefs (nplanes,ny,nx), smin, smax f32 already cut to the block; S has slin's dtype; exflag (ny,nx) u8 or NULL
   (|z| > 1 at the last evaluation: the reference's second return value).  Host arrays.  Exact. */
int rip_stage_invlinearity(rip_ctx *ctx, const void *slin, int dtype, int ny, int nx, int nplanes, const float *coefs,
                           const float *smin, const float *smax, void *S, uint8_t *exflag);

/* ---- noise layers ---------------------------------------------------- */
/* gen_noise_image.make_noise_cube, read-noise injection (gen_noise_image.py:120-134): per group k and active pixel
   data' = u16(rint(clip(f32(data) + f32(f64(normal) * (f64(read) / sqrt(f64(N_k)))), 0, 65535))); border pixels unchanged.
   cube, out (ngrp,ny,nx) u16; read_noise (ny,nx) f32; nreads[ngrp]; normals (ngrp,ny-2nb,nx-2nb) f32 standard normal
   deviates from the caller, or NULL: drawn on the device from a counter-based generator keyed by (seed, layer, group,
   pixel).  Host arrays.  Exact given the normals. */
int rip_stage_noise_inject(rip_ctx *ctx, const uint16_t *cube, int ngrp, int ny, int nx, int nb, const float *read_noise,
                           const int32_t *nreads, const float *normals, uint64_t seed, uint32_t layer, uint16_t *out);

/* sim_to_isim.noise_1f_frame (sim_to_isim.py:265-303): nframes frames of rows x width samples of 1/f noise (unit variance per
   logarithmic frequency range), the generator behind the correlated part of a read-noise layer (:376-399).  For each frame
   L = 2*rows*width complex samples (n_k + i n_{L+k}) |k|^-1/2 are Fourier transformed in f64 (hipFFT), the real part of the
   first L/2 outputs / sqrt(2) minus its mean is the frame, cast to f32.  normals (nframes, 2L) f64 standard normal deviates
   from the caller, or NULL: drawn on the device (seed, stream_id).  out (nframes, rows, width) f32.  Host arrays.  Agrees
   with the reference's numpy FFT to rounding (~1e-12 relative before the cast), not bit for bit. */
int rip_stage_noise_1f(rip_ctx *ctx, int rows, int width, int nframes, const double *normals, uint64_t seed,
                       uint32_t stream_id, float *out);

/* gen_noise_image.make_noise_cube, resampled Poisson layer 'P..r' (gen_noise_image.py:262-331) on n pixels: electrons per frame
   e = clip(skylevel * gain * frame_time, 0); for each read a Poisson deviate of mean e, re-centred, in DN, accumulated into the
   change of every resultant (group j = reads group_first[j] .. + group_count[j] - 1), then diff += sum_j w[endslice][j] *
   delta[j] with the weight vector of the ramp's end slice (weights (ngrp,ngrp) f32 row-major, has_weights[es] = 0 where the
   reference has no vector; endslice (n) i8 already mapped as the reference does, <= 0 -> ngrp - 1).  samples (nsamp,n) f64
   Poisson deviates from the caller, or NULL: drawn on the device (inversion / PTRS on Philox uniforms keyed by seed, layer,
   read, pixel).  gain (n) f32 or f64, already clipped to [1e-4, 1e4].  Up to 16 groups.  Host arrays; diff in/out.
   Exact given the deviates. */
int rip_stage_poisson_resample(rip_ctx *ctx, const float *skylevel, const void *gain, int gain_dtype, size_t n,
                               double frame_time, int ngrp, const int32_t *group_first, const int32_t *group_count,
                               const float *weights, const uint8_t *has_weights, const int8_t *endslice,
                               const double *samples, int nsamp, uint64_t seed, uint32_t layer, float *diff);

/* ---- statistics over many noise realisations of one ramp -- DEVICE pointers, asynchronous ------- */
/* Replaces the per-pixel arithmetic of validation_tests/many_realizations.py:57-106 on stacks (nseeds, rows, nx) that stay
   in HBM (256 realisations of a 4096 x 4096 SCA: 56 GB).  All results are exact (f32 operations in the reference's order,
   medians by selection). */

/* :69-72  out (ny,nx) f32 = f32(cube[ga]) - f32(cube[gb]) of a Level-1 cube (ngrp,ny,nx) u16 (the reference: ga = last
   group, gb = 1). */
int rip_stats_l1_diff(rip_ctx *ctx, const uint16_t *cube, int ngrp, int ny, int nx, int ga, int gb, float *out);

/* :74-77  one realisation's L2 planes into the stacks: image = slope and err = sqrt(err_read^2 + err_poisson^2) inside a
   zero border of nb pixels; good = 1 where the pixel passes the grown mask (grow[bit] as rip_stage_build_mask, host array)
   of the dq plane trimmed by nb (maskhandling.PixelMask1.build on the L2 file's dq), 0 elsewhere and in the border. */
int rip_stats_l2_pack(rip_ctx *ctx, const float *slope, const float *err_read, const float *err_poisson,
                      const uint32_t *pixeldq, int ny, int nx, int nb, const uint8_t grow[32], float *image, float *err,
                      uint8_t *good);

/* :78-101  the eight output planes for rows [y0, y0+nrows) of the (ny,nx) frame from stacks (nseeds,nrows,nx) holding every
   realisation in order; out (8,nrows,nx) f32 = ideal, median(diffs), median(images), N, mean, std, mean - ideal,
   median(err); N = mean = std = 0 in the border, mean = std = -1000 where N = 0 (:87).  alias_err != 0 reproduces the
   reference as written: its `images` and `err` stacks are memory maps of the same file (:58-59), so plane 2 equals
   plane 7.  ideal (nrows,nx) f32. */
int rip_stats_reduce(rip_ctx *ctx, int nseeds, const float *diffs, const float *images, const float *errs, const uint8_t *good,
                     const float *ideal, int y0, int nrows, int ny, int nx, int nb, int alias_err, float *out);

/* ---- measurement -------------------------------------------------------------------------- */
/* When enabled, rip_calibrate brackets each kernel group with HIP events on the ctx stream.
   rip_profile_read synchronises and returns the summed device time (ms) since the last read:
   out_ms[0] reference-pixel pre-pass, [1] cube stage (refpix apply + bias + linearity),
   [2] IPC, [3] ramp fit + finish; *ncalls = number of rip_calibrate calls summed. */
int rip_profile_enable(rip_ctx *ctx, int on);
int rip_profile_read(rip_ctx *ctx, double out_ms[4], int *ncalls);

/* options: "fused" (default 1) -- run the chain as the single fused kernel when the configuration allows it (complete chain on a
   u16 cube; f32 gain; 4, 9 or 11 Legendre planes; 6, 8 or 16 groups; flag words of the CALDIR set mergeable); 0 forces the
   stage-by-stage kernels.  Both give identical results. */
int rip_set_option(rip_ctx *ctx, const char *name, int value);
/* further options (results identical either way; they exist for A/B timing and tests):
   "chain2"  -- (default 1) the fused kernel (chain2_kernel.h: f32 or f64 ipc4d with 6, 8 or 16 groups); 0 = the stage kernels
                (rounds 1-2: a general fused kernel, dropped in round 3);
   "prepass_form" -- how the reference-pixel tables are made: -1 (default) by situation -- a pre-pass that overlaps the previous
                ramp's fused kernel as the nine small launches of refpix.hip (they slip into that kernel's tail), a pre-pass in
                front of its own ramp on the same stream as the single launch of refpix_one.hip (0.068 against 0.094 ms) where
                it covers the frame (up to 4096 rows, a reference output); 0 = refpix.hip always; 1 = refpix_one.hip wherever
                it covers the frame;
   "chain_reserve" -- (default 8) workgroup slots the 256-column fused kernel's grid leaves free; "chain_quad" -- (default 1) a
                last strip of at most 64 live columns is covered by workgroups whose four wave columns take a row range each
                (4096 x 4096: 504 workgroups of 139 steps instead of 510 of 143; timing-neutral, profiles/r04_summary.md);
   "pink_form" -- the complex-to-real transform of the 1/f frames: -1 (default) the library's own two-pass transform for
                power-of-two frame lengths (2^8 .. 2^21 points; csrc/pink_fft.h) and hipFFT otherwise, 0 hipFFT for every length;
   "overlap" -- run the reference-pixel pre-pass of a ramp on a second stream so that it overlaps the previous ramp's
                fused kernel: -1 (default) by situation -- wherever the fused kernel leaves room on the CUs (every form except the
                f64-ipc4d one of up to 8 groups, whose partial coefficient ring fills the LDS), 0 never, 1 always. */

/* how the last rip_calibrate ran: 0 = stage kernels, 2 = the fused kernel (1 and 3 were the general and the wave-private fused
   kernels of rounds 1-2) */
int rip_last_chain_form(rip_ctx *ctx);

/* pseudo-Poisson noise layers ("O" directives, gen_noise_image.py:173-240): per element of I (n doubles, host memory) the
   member of the Pearson family with the moments tilnu_21 I, tilnu_31 I, 3 tilnu_21^2 I^2 + tilnu_41 I, replacing
   L1_to_L2/GalPoisson/draw_with_tilnus.py:draw_from_Pearson (:12-135) and the solvers / samplers it calls.
   types  (n int32, or NULL): 0 = outside the admissible region (deviate 0), 1, 3, 4, 5, 6 = Pearson type;
   params (4 n doubles, or NULL): type 1: a, b, mean, c; 3: shape, scale, shift, sign; 4: m, nu, a, lambda; 5: a, b, mu, sign;
          6: alpha, beta, scale, shift -- the reference's formulas in f64 (pinned by goldens);
   draws  (n doubles, or NULL): one deviate each from a counter-based generator keyed by (seed, stream, element); the
          reference's scipy / numpy streams cannot be reproduced: parity of the random part unpinned (moments tested). */
int rip_stage_pearson(rip_ctx *ctx, size_t n, const double *I, double tilnu21, double tilnu31, double tilnu41, uint64_t seed,
                      uint32_t stream, double *draws, int32_t *types, double *params);

/* ---- Level-1 synthesis -- DEVICE pointers, asynchronous on the context's stream ------------------ */
/* The per-pixel work of from_sim/sim_to_isim.py that turns an electron-count image into a raw exposure: make_l1_fullcal
   (:163-262, with the loop of romanisim.l1.apportion_counts_to_resultants it calls), fill_in_refdata_and_1f (:306-403) and the
   EXTRACT_REF block (:711-730).  Calibration arrays as the CALDIR files hold them, resident in HBM: */
typedef struct rip_synth_cal {
    int32_t ny, nx, nb;        /* full frame and its reference-pixel border; active region (ny-2nb, nx-2nb)             */
    int32_t channelwidth;      /* columns per readout channel (nx / channelwidth channels: 32 of 128 on a full frame);   */
                               /* the reference output has one channel's width                                           */
    int32_t nplanes;           /* Legendre planes of the linearity file, 2..17                                          */
    int32_t gain_dtype;        /* RIP_F32 or RIP_F64                                                                     */
    int32_t ipc_dtype;         /* RIP_F32 or RIP_F64                                                                     */
    int32_t amp33_valid;       /* read file has a valid amp33 block (med, std, M_PINK, RU_PINK)                          */
    const void *gain;          /* (ny,nx)                                                                               */
    const float *read_noise;   /* read file "data" (ny,nx)                                                              */
    const float *resetnoise;   /* read file "resetnoise" (ny,nx)                                                        */
    const float *dark_slope;   /* dark file "dark_slope" (ny,nx)                                                        */
    const float *dark;         /* the LAST ngrp planes of the dark file's "data": (ngrp,ny,nx)                          */
    const float *lin_coefs;    /* (nplanes,ny,nx)                                                                       */
    const float *smin, *smax;  /* (ny,nx)                                                                               */
    const void *ipc4d;         /* (3,3,ny-2nb,nx-2nb), or NULL: no IPC (IL(..., ipc_file=None))                          */
    const float *biascorr;     /* the last ngrp planes of the biascorr file's "data": (ngrp,ny-2nb,nx-2nb), or NULL      */
    double tbias;              /* biascorr "t0" (used only with biascorr)                                               */
    const float *amp33_med, *amp33_std;   /* (ny,channelwidth), or NULL                                                 */
    double m_pink, ru_pink;    /* amp33 "M_PINK", "RU_PINK"                                                             */
    double u_pink, c_pink;     /* read file anc "U_PINK", "C_PINK"                                                      */
} rip_synth_cal;

/* romanisim.l1.apportion_counts_to_resultants, the sampling part (dependency absent from the reference tree: published
   algorithm restated, oracle/l1sim.py): read r takes Binomial(counts - collected so far, (t_r - t_{r-1}) / (t_last - t_{r-1}))
   electrons; reads_e (nreads,nya,nxa) i32 = electrons collected up to each read, so reads_e[nreads-1] == counts.  counts
   (nya,nxa) f32 holds integers, or -- poisson != 0 -- the MEAN, of which a Poisson deviate is drawn first (what
   Image2D.simulate :660-662 adds before it calls make_l1_fullcal).  t_reads: nreads times, HOST array, ascending.
   Deviates from the device generator (Philox; inversion / BTRS binomial, inversion / PTRS Poisson) keyed by (seed, read,
   pixel).  The distribution is what is reproduced, not romanisim's numpy stream.  Asynchronous like the other rip_synth_*
   entries when t_reads equals those of the previous call (the device copy of the share table is kept); a NEW table first
   waits for the work queued on the stream. */
int rip_synth_apportion(rip_ctx *ctx, const float *counts, int nya, int nxa, int poisson, int nreads, const double *t_reads,
                        uint64_t seed, int32_t *reads_e);

/* make_l1_fullcal :203-260 after the sampling: reset noise in electrons (normal * resetnoise * gain - t0 * dark_slope / gain),
   per read IL.apply(electrons + reset, electrons=True) (ipc_linearity.py:461-513: i32 + f32 -> f64, ipc_fwd, / gain, 24
   bisection steps of invlinearity in f64), resultant = f32 mean of its reads, + normal * read_noise / sqrt(reads), + biascorr,
   rounded half to even.  group_count: HOST array, reads per resultant (sum = the number of planes of reads_e).
   normals_reset (nya,nxa) f32 and normals_read (ngrp,nya,nxa) f32 standard normal deviates, or NULL: device generator (seed).
   Outputs, each optional: start_e (nya,nxa) f32 the reset-noise image; resultants (ngrp,nya,nxa) f32; cube (ngrp,ny,nx) u16 --
   the resultants clipped to 0..65535 inside a zero border (what romanisim.l1.make_asdf builds around them).
   Exact given the deviates (goldens from the reference's function). */
int rip_synth_resultants(rip_ctx *ctx, const rip_synth_cal *cal, int ngrp, const int32_t *group_count, const int32_t *reads_e,
                         const float *normals_reset, const float *normals_read, uint64_t seed, float *start_e,
                         float *resultants, uint16_t *cube);

/* fill_in_refdata_and_1f :306-403, in place on cube (ngrp,ny,nx) u16 and amp33 (ngrp,ny,channelwidth) u16 (or NULL: no
   reference output): reference pixels = normal * read / sqrt(reads) + normal * resetnoise + dark, active pixels kept, every
   pixel + (1/f frame of its channel * U_PINK + common frame * C_PINK) / sqrt(reads) with odd channels mirrored, rounded and
   clipped to u16; amp33 = med + (normal * std + RU_PINK * frame + M_PINK * common) / sqrt(reads), cast.  banding == 0 skips
   the correlated noise (fill_in_banding=False; amp33 is then left untouched, as in the reference).
   normals (ngrp+1,ny,nx) f32, frames (ngrp,nch+2,ny,channelwidth) f32 in the reference's draw order per group (common, channels
   0..nch-1, reference output; nch = nx / channelwidth) and white33 (ngrp,ny,channelwidth) f32, or NULL each: device generators (seed; frames as
   rip_stage_noise_1f makes them).  Exact given the deviates (goldens from the reference's function). */
int rip_synth_fill(rip_ctx *ctx, const rip_synth_cal *cal, int ngrp, const int32_t *group_count, int banding, const float *normals,
                   const float *frames, const float *white33, uint64_t seed, uint16_t *cube, uint16_t *amp33);

/* rip_stage_noise_1f with the frames left on the device: out (nframes,rows,width) f32 DEVICE memory. */
int rip_synth_noise_1f(rip_ctx *ctx, int rows, int width, int nframes, uint64_t seed, uint32_t stream_id, float *out);
/* The 1/f frames of the NEXT rip_synth_fill(frames = NULL, banding != 0, the same seed and geometry: rows = ny, width =
   channelwidth, nframes = ngrp * (nx / channelwidth + 2)) made AHEAD on the context's second stream, so that the Fourier
   transforms (HBM-bound) run beside the inverse-linearity kernel (arithmetic-bound) queued on rip_stream() between this call
   and the fill: same frames, same bits.  They start behind whatever rip_stream() holds at the time of the call -- call it AFTER
   rip_synth_apportion (HBM-bound too) and before rip_synth_resultants.  The fill waits for them; a fill with another seed or
   geometry, or any other 1/f call, waits too and makes its own.  No-op without a second stream. */
int rip_synth_frames_ahead(rip_ctx *ctx, int rows, int width, int nframes, uint64_t seed);

/* EXTRACT_REF (:711-730) on n-element planes: reference_read = data[0]; data[k] = clip(i32(data[k]) - (i32(data[0]) -
   offset), 0, 65535) for k = 1..ngrp-1, in place (the caller drops plane 0).  Used for the cube and for amp33.  Exact. */
int rip_synth_extract_ref(rip_ctx *ctx, uint16_t *data, int ngrp, size_t n, int offset, uint16_t *reference_read);

/* ---- diagnostics ------------------------------------------------------------------------- */
/* floating-point options of a context.  "guard_band": relative half-width of the band around the jump
   threshold inside which the significance is re-evaluated in the reference's exact operation order
   (default 1e-5; INFINITY = always exact).  Results do not depend on it unless it is set below ~1e-6; it
   exists so that tests can force either path.  Per context (round 1 had a process-wide setter). */
int rip_set_option_f64(rip_ctx *ctx, const char *name, double value);

#ifdef __cplusplus
}
#endif
#endif /* ROMANHIP_H */
