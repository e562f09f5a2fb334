/*
 * mcpm.h -- C ABI of the MI355X-native particle-mesh (PM) forward model + hand-written adjoint.
 *
 * Drop-in boundary for the hot path of hsimonfroy/montecosmo.  The reference has no FFI: its
 * boundary is the Python function surface of montecosmo/nbody.py (consumed by bricks.py:10 and
 * model.py:23-25).  Each entry point below names the reference function it replaces
 * (file:line relative to the reference checkout); `montecosmo_amd/nbody.py` binds them with ctypes
 * and re-exports the reference's names.  INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - extern "C", POD only.  Every `*_d` / unnamed array pointer is a DEVICE pointer owned by the
 *     caller; the library owns only the plan (rocFFT plans, scratch meshes, outlier list).
 *   - Every function returns 0 (MCPM_OK) or a negative MCPM_E_* code; mcpm_last_error() gives text.
 *     No C++ exception crosses the ABI, nothing aborts.
 *   - Work is enqueued on the plan's HIP stream and the call returns without synchronising.
 *   - Meshes are C-order [x][y][z] float32 (z fastest); half-spectra are [x][y][z/2+1] interleaved
 *     complex64 -- the layout of jnp.fft.rfftn (nbody.py:589).  nz must be even (utils.py:769-776).
 *   - Particles are AoS float32 [N][3].  Positions are in cell units of the mesh, periodic, any real
 *     value with |pos| < 32767 (the reference does its index arithmetic in int16, nbody.py:369).
 *   - FFTs are unnormalised in both directions (rocFFT); the 1/M of numpy/jax `irfftn` is folded into
 *     the `scale` argument of the k-space kernels.
 */
#ifndef MCPM_H
#define MCPM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcpm_plan mcpm_plan;

#define MCPM_OK 0
#define MCPM_E_SHAPE (-1)       /* bad mesh / lattice shape */
#define MCPM_E_ORDER (-2)       /* unsupported assignment or finite-difference order */
#define MCPM_E_HIP (-3)         /* HIP runtime error */
#define MCPM_E_ROCFFT (-4)      /* rocFFT error */
#define MCPM_E_RCCL (-5)        /* RCCL error */
#define MCPM_E_ARG (-6)         /* null pointer / bad argument */
#define MCPM_E_NOMEM (-7)       /* device allocation failed */
#define MCPM_E_UNSUPPORTED (-8) /* valid request not implemented on this path */

/* Position encodings of a float32 [N][3] particle array. */
#define MCPM_POS_ABSOLUTE 0 /* absolute cell coordinates */
#define MCPM_POS_LATTICE 1  /* displacement from the plan's particle lattice point of particle i, the
                               i-th row of regular_pos(mesh_shape, ptcl_shape) (bricks.py:593-603);
                               requires n == px*py*pz.  Keeps fp32 precision at large meshes. */

/* Finite-difference order selectors for the k-space kernels (nbody.py:109-163). */
#define MCPM_FD_INF 0
#define MCPM_FD_2 2
#define MCPM_FD_4 4

/* ---- plan ------------------------------------------------------------------------------------ */
/* mesh (nx,ny,nz), particle lattice (px,py,pz), HIP stream (hipStream_t, may be NULL). */
int mcpm_plan_create(int nx, int ny, int nz, int px, int py, int pz, void *stream, mcpm_plan **plan);
/* x-slab plan for rank `rank` of `nranks`: (nx,ny,nz) is the GLOBAL mesh; this rank owns mesh
   planes [rank*nx/nranks, (rank+1)*nx/nranks) and the lattice particles of those planes (Lagrangian ownership:
   particles never migrate).  Particle kernels then work on the ghost-extended local mesh
   (nx/nranks + 2*ghost, ny, nz), non-periodic in x; ghost planes are exchanged by the host (montecosmo_amd/dist.py). */
int mcpm_plan_create_slab(int nx, int ny, int nz, int nranks, int rank, int ghost, void *stream, mcpm_plan **plan);
/* Slab plans: cumulative number of particle deposits that fell beyond the ghost planes (clamped to the edge); a
   non-zero count means `ghost` is too small for the displacements (host sync). */
int mcpm_plan_slab_oob(mcpm_plan *plan, int64_t *count);
int mcpm_plan_destroy(mcpm_plan *plan);
const char *mcpm_last_error(const mcpm_plan *plan); /* plan may be NULL: last error of a failed create */
/* ABI revision string; the Python loader (montecosmo_amd/_lib.py) refuses a library that reports another one. */
#define MCPM_ABI_VERSION "mcpm 0.6 (gfx950)"
const char *mcpm_version(void);
/* Tiled CIC paints (montecosmo_amd/csrc/paint_tiled.hip).  A tile's window is a box of lattice points per axis -- chosen on the device
   for every input and every tile from the displacement field around it, or (16 + 2 halo + 1)^3 around the tile's bulk displacement when a
   halo is fixed (mcpm_plan_set_halo) or the mesh has fewer than 2048 tiles; what a window misses travels through per-tile
   buckets (integer LDS sums, so the paint stays bitwise reproducible).  mcpm_plan_last_bucketed: (particle, tile) pairs the last tiled paint routed through
   the buckets.  mcpm_plan_last_outliers: particles / pairs it had to deposit with f32 global atomics instead (non-finite or
   absurd displacements, bucket overflow).  Both synchronise the host. */
int mcpm_plan_last_outliers(mcpm_plan *plan, int64_t *count);
int mcpm_plan_last_bucketed(mcpm_plan *plan, int64_t *count);
/* Everything the last tiled paint counted, in one call (synchronises the host): out13[0..6] = wild particles, wild + overflow pairs,
   slab deposits beyond the ghost planes (cumulative), appends that found their bucket full, tiles with a non-empty bucket, bucketed
   pairs, suspects (particles handed to the exact coverage test); out13[7] = window points of all tiles (divided by the particle count:
   window visits per particle); out13[8 + h], h = 0..4 = number of tiles whose widest window axis spans what a symmetric halo of h would
   (17 + 2h - 1 or 17 + 2h points; all zero when the plan's windows are not centred). */
int mcpm_plan_last_paint_stats(mcpm_plan *plan, int64_t *out13);
/* Tuning knobs: halo radius (1, 2, 3, 4 or 6 cells: a symmetric window around the tile's bulk displacement; 0 = the default: a window BOX per tile
   and per axis chosen ON THE DEVICE for every input from the sampled displacement ranges of the Lagrangian blocks around the tile -- same input,
   same windows, so results stay bitwise reproducible; meshes below 2048 tiles: halo 4 up to 2^24 cells, 3 above) and whether windows are centred on the local bulk
   displacement (default 1; 0 = on the tile itself, which needs halo 4 at the benchmark's 2-cell rms displacement). */
int mcpm_plan_set_halo(mcpm_plan *plan, int halo);
int mcpm_plan_set_centre(mcpm_plan *plan, int centre);
/* Thread -> particle map of the lattice-mode particle kernels (read, kick+drift, adjoint step, ...): 1 (default where px, py are
   even and pz is a multiple of 64; MCPM_LATTICE_PATCH=0 turns it off) = the four waves of a workgroup take a 2 x 2 (x, y) patch
   of lattice rows, 64 consecutive z each, so that the mesh rows their CIC stencils share (9 instead of 16) are fetched into
   the CU's L1 once; 0 = 256 consecutive z of one row.  A permutation of the work: results do not depend on it.  A/B knob. */
int mcpm_plan_set_lattice_patch(mcpm_plan *plan, int on);
/* Accumulator of the tiled three-component paint (mcpm_paint3_f32, the adjoint of the force read): 1 = fixed point
   (default: 32-bit fields in 64-bit integer LDS atomics, exact order-independent sums, overflow proven per tile by a
   bound field, flagged tiles repainted in f64; particles.hip), 0 = f64 tiles.  mcpm_plan_last_redo returns how many
   tiles the last fixed-point paint handed to the f64 kernel (host sync). */
int mcpm_plan_set_paint3_fixed(mcpm_plan *plan, int fixed);
int mcpm_plan_last_redo(mcpm_plan *plan, int64_t *count);

/* Optional per-stage profile: HIP events on the plan's stream around every leaf stage (paint, FFTs, k-space,
   read, fused particle kernels).  mcpm_plan_profile_read synchronises, fills up to nmax entries of
   accumulated milliseconds / algorithmic bytes (SURVEY.md 8d accounting) / launch counts per stage, resets the
   record and returns the number of stages (negative on error).  mcpm_stage_name(i) names stage i. */
int mcpm_plan_profile(mcpm_plan *plan, int enable);
int mcpm_plan_profile_read(mcpm_plan *plan, int nmax, double *ms, double *bytes, int64_t *calls);
const char *mcpm_stage_name(int stage);

/* ---- FFT (replaces jnp.fft.rfftn / irfftn at nbody.py:589,603,620,627,630) ------------------ */
int mcpm_fft_r2c(mcpm_plan *plan, const float *real, float *spec, int batch);
/* Unnormalised (M x irfftn); `spec` is destroyed. */
int mcpm_fft_c2r(mcpm_plan *plan, float *spec, float *real, int batch);

/* ---- mass assignment (nbody.py:365-427) ----------------------------------------------------- */
/* wrap(id0) of nbody.py:372-375 for every particle, int16 [N][3]: the integer part of the path. */
int mcpm_cell_index(mcpm_plan *plan, const float *pos, int64_t n, int pos_mode, int order, int16_t *idx);
/* paint (nbody.py:365-396): mesh (+)= sum_p w_p prod_a K(c_a - x_pa).  weights may be NULL (then
   wscalar is used); element p is weights[p*wstride].  order 1 (NGP) or 2 (CIC).
   accumulate = 0 overwrites mesh, 1 adds to it. */
int mcpm_paint_f32(mcpm_plan *plan, const float *pos, int64_t n, int pos_mode, const float *weights,
                   int64_t wstride, float wscalar, int order, float *mesh, int accumulate);
/* Three weighted paints in one pass: meshes3[c] (+)= paint(pos, weights = weights3[:, c]), c = 0..2 (the VJP of a
   three-component read w.r.t. its meshes).  weights3 is float32 [N][3]; the three meshes are contiguous. */
int mcpm_paint3_f32(mcpm_plan *plan, const float *pos, int64_t n, int pos_mode, const float *weights3, int order,
                    float *meshes3, int accumulate);
/* read (nbody.py:398-427) of `ncomp` contiguous meshes at once: out[p*ncomp + c]. */
int mcpm_read_f32(mcpm_plan *plan, const float *pos, int64_t n, int pos_mode, const float *meshes,
                  int ncomp, int order, float *out);
/* VJP of paint w.r.t. pos and weights: pos_bar[N][3] (overwritten), weights_bar[N] (may be NULL). */
int mcpm_paint_vjp_f32(mcpm_plan *plan, const float *pos, int64_t n, int pos_mode, const float *weights,
                       int64_t wstride, float wscalar, int order, const float *mesh_bar, float *pos_bar,
                       float *weights_bar);
/* VJP of read w.r.t. pos: pos_bar[N][3] = sum_c out_bar[p][c] * d read_c / d pos (overwritten).
   (The VJP w.r.t. the mesh is mcpm_paint_f32 with weights = out_bar[:, c].) */
int mcpm_read_vjp_pos_f32(mcpm_plan *plan, const float *pos, int64_t n, int pos_mode, const float *meshes,
                          int ncomp, int order, const float *out_bar, float *pos_bar);

/* ---- k-space kernels (nbody.py:109-163, :596-603, :611-627) --------------------------------- */
/* out[c] = scale * (-(i k_c)) * invlaplace(k) * [gaussian(kcut)] * [1/sinc^(2*deconv_order)] * in,
   c = 0..2, three contiguous half-spectra (pm_forces, nbody.py:596-603).  kcut <= 0 means inf. */
int mcpm_kspace_force_f32(mcpm_plan *plan, const float *spec_in, float *spec_out3, float scale,
                          int lap_fd, int grad_fd, float kcut, int deconv_order);
/* Adjoint: out = scale * sum_c conj(multiplier_c) * in3[c].  zweights=1 multiplies by the irfftn
   multiplicity (1,2,..,2,1) along kz (cotangent of a half-spectrum returned to the caller);
   hermitian=1 applies the Hermitian projection on the kz=0 / Nyquist planes (spectrum fed to a C2R). */
int mcpm_kspace_force_vjp_f32(mcpm_plan *plan, const float *spec_in3, float *spec_out, float scale,
                              int lap_fd, int grad_fd, float kcut, int deconv_order, int zweights,
                              int hermitian, int accumulate);
/* out[ab] = scale * (i k_a)(i k_b) * invlaplace(k) * in for ab = 00,01,02,11,12,22 (nbody.py:611-627). */
int mcpm_kspace_hessian_f32(mcpm_plan *plan, const float *spec_in, float *spec_out6, float scale,
                            int lap_fd, int grad_fd);
int mcpm_kspace_hessian_vjp_f32(mcpm_plan *plan, const float *spec_in6, float *spec_out, float scale,
                                int lap_fd, int grad_fd, int zweights, int accumulate);
/* Elementwise half-spectrum operator of the observation-side painting (next row):
   out (+)= scale * exp(i shift (kx+ky+kz)) / prod_a sinc(k_a/2pi)^deconv_order * in  -- the interlacing phase of
   `interlace` (nbody.py:525) and the kernel deconvolution of `deconv_paint` (nbody.py:315-334).  conj = 1 conjugates
   the phase and inv_zweights = 1 divides by the irfftn multiplicity (1,2,..,2,1): together the adjoint fed to a C2R. */
int mcpm_kspace_phase_f32(mcpm_plan *plan, const float *in, float *out, float scale, float shift, int deconv_order,
                          int conj, int inv_zweights, int accumulate);
/* delta2 = sum_{i<j} h_ii h_jj - h_ij^2 from six contiguous real meshes (nbody.py:615-627). */
int mcpm_hessian_combine_f32(mcpm_plan *plan, const float *hess6, float *delta2);
int mcpm_hessian_combine_vjp_f32(mcpm_plan *plan, const float *hess6, const float *delta2_bar, float *hess6_bar);

/* Poisson solve on meshes (nbody.py:589, :596-603 with fd_order = inf): three real force meshes
   irfftn(-(i k_c)(-1/k^2) rfftn(rho)) from a real density mesh, and the adjoint (three real cotangent meshes ->
   rho_bar).  Power-of-two meshes run the hand-written five-pass FFT with the k-space multiply fused into the
   x pass (fftpm.hip); other sizes run rocFFT + the k-space kernels.  fm3 / fbar3: three contiguous meshes. */
int mcpm_force_meshes_f32(mcpm_plan *plan, const float *rho, float *fm3);
int mcpm_force_meshes_vjp_f32(mcpm_plan *plan, const float *fbar3, float *rho_bar);

/* Pass-level entry points of the same solve for slab plans: the host (montecosmo_amd/dist.py) places an all-to-all
   between the y and the x pass.  Spectra are complex64 in the internal padded layout, mcpm_slab_spec_elems()
   complex per spectrum = (nx/ranks) * ny * (nz/2 + 16).
     zfwd  : `batch` real meshes of nx/ranks planes (mesh b at real + b*real_bstride floats) -> `batch` spectra
     ycol  : FFT along y (sign -1 forward / +1 inverse) of `batch` spectra; *_packed = transposed-order layout
             [c][dest rank][x_local][y_local][nzp] (what one all-to-all per spectrum exchanges), plain =
             [c][x_local][y][nzp]
     xfused: mode 0: one spectrum [x][y_local][nzp] -> x FFT -> {-i kx L X, -i L X} (L = -1/(M k^2)) -> inverse x FFT
             -> TWO spectra A, G in [c][dest rank][x_local][y_local][nzp]; mode 1 is the adjoint (two -> one).
     ycol2 : the y pass that carries the remaining force factors, so only two spectra cross the all-to-all:
             expand = 1 (inverse): {A, G} -> three force spectra {IFFTy A, IFFTy(ky G), kz IFFTy G};
             expand = 0 (forward, adjoint): three spectra {a, b, c} -> {FFTy a, ky FFTy b + kz FFTy c}.
             Modes 2..5 are the spectrum-side variants of lpt: the single spectrum is the caller's FULL plain
             half-spectrum [nx][ny][nz/2+1] (this rank touches its y rows only) and one x transform disappears:
             2: spectrum -> 3 force spectra, 3: spectrum -> 6 Hessian spectra (00 01 02 11 12 22),
             4: 3 -> spectrum cotangent (overwrites the rank's rows), 5: 6 -> spectrum cotangent (accumulates)
     zinv  : `batch` spectra -> `batch` real meshes (unnormalised; the 1/M sits in xfused). */
int64_t mcpm_slab_spec_elems(const mcpm_plan *plan);
int mcpm_slab_zfwd(mcpm_plan *plan, const float *real, int64_t real_bstride, float *spec, int batch);
int mcpm_slab_ycol(mcpm_plan *plan, const float *in, float *out, int batch, int sign, int in_packed, int out_packed);
/* parts: bit 0 = the spectrum-0 transform (A / a), bit 1 = the other two; 3 = all.  The halves travel in separate
   all-to-alls, so a caller can run one half while the other is still in flight. */
int mcpm_slab_ycol2(mcpm_plan *plan, const float *in, float *out, int expand, int in_packed, int out_packed, int parts);
/* Restricts the following zfwd / zinv / ycol / ycol2 calls to the local planes [x0, x0 + count) (pointer arguments
   still address plane 0).  Planes are independent in those passes, so a caller can transform the planes a halo
   exchange does not touch while the exchange is in flight.  (0, nx_local) restores the default. */
int mcpm_slab_set_window(mcpm_plan *plan, int x0, int count);
/* Splits every all-to-all (transposed-order) layout into `chunks` chunks of nx_local / chunks planes:
   [c][chunk][rank][x in chunk][y_local][nzp] instead of [c][rank][x_local][y_local][nzp], so that a transpose becomes
   `chunks` all-to-alls of contiguous regions (each 1/chunks of a spectrum) that the caller overlaps with the z / y passes
   of the other chunks (planes of a chunk: mcpm_slab_set_window).  chunks: a power of two dividing nx_local; 1 = default.
   Applies to the buffers of the following mcpm_slab_* calls; set it once after mcpm_plan_create_slab. */
int mcpm_slab_set_chunks(mcpm_plan *plan, int chunks);
int mcpm_slab_xfused(mcpm_plan *plan, const float *in, float *out, int mode);
int mcpm_slab_zinv(mcpm_plan *plan, const float *spec, float *real, int64_t real_bstride, int batch);
/* z C2R of three spectra (spec_elems apart) into ONE interleaved real mesh [x][y][z][3] (window of local planes as above). */
int mcpm_slab_zinv3_il(mcpm_plan *plan, const float *spec3, float *real_il);

/* ---- One slab-decomposed BullFrog step and its adjoint behind the ABI ("the library
   owns ... the RCCL communicator").  The library issues the FFT transposes (all-to-all) and the ghost-plane exchanges
   (point-to-point with the two x neighbours) on its own communication stream, ordered against the plan's stream with
   events: one host call per step.  No reference counterpart (montecosmo/script.py:13-20 runs independent chains); the
   kernels, windows and their order are those of montecosmo_amd/dist.py SlabPM.step_gen / step_vjp_gen, so both paths give
   bitwise equal results (tests/test_dist.py).

   Transport, chosen once per plan:
     mcpm_slab_comm_init_local  one rank: ghost exchanges are device copies, the transpose is the identity;
     mcpm_slab_comm_init_rccl   RCCL communicator owned by the plan.  `id128`: the 128-byte ncclUniqueId that rank 0 obtained
                                from mcpm_slab_rccl_unique_id and the host distributed to every rank.  librccl is opened
                                at run time (the copy already mapped into the process, else ROCm's); MCPM_E_RCCL on failure;
     mcpm_slab_comm_init_ops    host-provided callbacks (tests drive the multi-rank algebra through gloo with them).
   mcpm_slab_bind_workspace: device scratch the steps work in (caller-allocated, as everywhere in this ABI):
     rho (nxe ny nz), f3 (3 nxe ny nz), s1a, s1b (mcpm_slab_spec_elems() complex each), s6a, s6b (6 spectra each),
     Fb (3 N_local), halo (6 ghost ny nz floats), nxe = nx/nranks + 2 ghost. */
typedef struct mcpm_comm_ops {
    void *ctx;
    /* Start n_send sends and n_recv receives (device pointers, byte counts, peer ranks), ordered after the work already
       enqueued on `stream`; between one pair of ranks the k-th send meets the k-th receive.  *ticket identifies the batch. */
    int (*p2p_begin)(void *ctx, int n_send, const void *const *send_ptrs, const int64_t *send_bytes, const int *send_peers,
                     int n_recv, void *const *recv_ptrs, const int64_t *recv_bytes, const int *recv_peers, void *stream,
                     int *ticket);
    /* Order `stream` after the batch `ticket`. */
    int (*wait)(void *ctx, int ticket, void *stream);
    /* Replace the device float by its maximum over ranks, in stream order. */
    int (*allreduce_max_f32)(void *ctx, float *dev_value, void *stream);
} mcpm_comm_ops;
/* One small round of every exchange pattern of a slab step (equal-split all-to-all, both neighbour exchanges, max all-reduce) on
   the plan's transport, verified on the host: MCPM_OK, or MCPM_E_RCCL with mcpm_last_error saying which pattern failed.
   Collective over the plan's ranks; synchronises the plan's stream.  dist.SlabPM runs it once after bringing the communicator up
   and, if any rank fails, keeps issuing the exchanges through torch.distributed instead (reference: there is none -- the
   reference shards nothing, script.py:13-20 runs independent chains). */
int mcpm_slab_comm_selftest(mcpm_plan *plan);
/* Gives the plan's transport up (communicator, communication stream, events) after draining the plan's stream; the plan itself
   stays usable, and mcpm_slab_comm_init_* may be called again.  mcpm_plan_destroy does this too. */
int mcpm_slab_comm_shutdown(mcpm_plan *plan);
int mcpm_slab_rccl_unique_id(void *id128);
int mcpm_slab_comm_init_local(mcpm_plan *plan);
int mcpm_slab_comm_init_rccl(mcpm_plan *plan, const void *id128);
int mcpm_slab_comm_init_ops(mcpm_plan *plan, const mcpm_comm_ops *ops);
int mcpm_slab_bind_workspace(mcpm_plan *plan, float *rho, float *f3, float *s1a, float *s1b, float *s6a, float *s6b, float *Fb,
                             float *halo);
/* One DKD step on this rank's slab (nbody.py:933-951 as merged in DESIGN.md "Stepping form"): paint -> ghost add -> slab
   Poisson solve -> ghost fill -> read+kick+drift.  depth: ghost planes this step's displacements can reach (1..ghost).
   f3_out (nxe, ny, nz, 3) receives the interleaved force mesh (the adjoint's checkpoint).  Also enqueues the measurement of
   max |d_x| over ranks of x_out (mcpm_plan_track_dmax slots; mcpm_slab_dmax_read). */
int mcpm_slab_step_f32(mcpm_plan *plan, const float *x, const float *v, double alpha, double beta, double tau, int paint_order,
                       int depth, float *f3_out, float *x_out, float *v_out);
/* Adjoint of mcpm_slab_step_f32 (f3: its force mesh): xb, vb updated in place; scalar cotangents accumulated as in
   mcpm_step_adjoint_particles_f32.  has_next: the adjoint of the step with (next_beta, next_tau) follows (its force
   cotangent is then written by this call's particle kernel, mcpm_plan_hint_next_adjoint). */
int mcpm_slab_step_vjp_f32(mcpm_plan *plan, const float *x, const float *v, const float *f3, double alpha, double beta,
                           double tau, int paint_order, int depth, float *xb, float *vb, double *alpha_bar, double *beta_bar,
                           double dtau_ddg, double *dg_bar, int has_next, double next_beta, double next_tau);
/* Ghost-depth measurements: every mcpm_slab_step_f32 enqueues max |d_x| over ranks of its x_out into a ring of four pinned
   floats; mcpm_slab_dmax_seq = measurements enqueued so far (the next step's measurement gets that number);
   mcpm_slab_dmax_read waits for measurement `seq` and returns it (*valid = 0: never taken or already overwritten).  Read one
   step late (montecosmo_amd/dist.py) the wait is free and the host never stops for the depth. */
int64_t mcpm_slab_dmax_seq(const mcpm_plan *plan);
int mcpm_slab_dmax_read(mcpm_plan *plan, int64_t seq, float *value, int *valid);
/* Self-test of the library's hand-written 12-byte streaming store (global_store_dwordx3 ... nt by inline asm, whose store-data
   hazard the compiler cannot pad: csrc/mcpm_internal.h MCPM_STORE_DATA_HAZARD_NOP).  Writes record i = ((3i, 3i+1, 3i+2) mod 2^24)
   as floats into out[3n] on `stream`.  mode 0: plain compiler store (the twin to compare with); 1: the asm store followed at once
   by VALU writes of its three data registers (must equal mode 0); 2: the same WITHOUT the wait states (informational only: shows
   whether the part exhibits the hazard); 3: the library's own store helper followed by VALU writes of its sources.  No reference
   counterpart (test infrastructure of the boundary). */
int mcpm_selftest_store3_nt(void *stream, float *out, int64_t n, int mode);
/* out = a x + b y over n floats (the ghost-plane additions of the slab steps; also the drift / kick building block). */
int mcpm_axpby_f32(mcpm_plan *plan, const float *x, const float *y, int64_t n, float a, float b, float *out);

/* ---- forces (nbody.py:583-631) -------------------------------------------------------------- */
/* pm_forces with mesh = shape tuple: paint -> R2C -> k-space -> 3 C2R -> read; forces[N][3].
   Leaves the three force meshes in the plan (mcpm_plan_force_meshes). */
int mcpm_pm_forces_f32(mcpm_plan *plan, const float *pos, int64_t n, int pos_mode, int order,
                       int paint_deconv, int lap_fd, int grad_fd, float kcut, float *forces);
/* pm_forces with mesh = half-spectrum (not modified). */
int mcpm_pm_forces_spec_f32(mcpm_plan *plan, const float *spec, const float *pos, int64_t n, int pos_mode,
                            int order, int lap_fd, int grad_fd, float kcut, float *forces);
/* VJP of pm_forces (spectral kernels, no deconvolution / smoothing).  spec == NULL: painted case (mesh = shape
   tuple), pos_bar [N][3] carries the read and the paint dependence.  spec != NULL: pos_bar from the read, and
   spec_bar = cotangent of the half-spectrum (real-pair convention, irfftn multiplicity weights included). */
int mcpm_pm_forces_vjp_f32(mcpm_plan *plan, const float *spec, const float *pos, int64_t n, int pos_mode, int order,
                           const float *forces_bar, float *pos_bar, float *spec_bar);
/* The painted case with the options of nbody.py:583-604: finite-difference Laplace / gradient kernels (:125-163) and
   deconvolution of the painted density (:590-593); pos_bar carries the read and the paint dependence. */
int mcpm_pm_forces_vjp_opts_f32(mcpm_plan *plan, const float *pos, int64_t n, int pos_mode, int order, int paint_deconv,
                                int lap_fd, int grad_fd, const float *forces_bar, float *pos_bar);
/* pm_forces2 (2LPT source, nbody.py:607-631). */
int mcpm_pm_forces2_f32(mcpm_plan *plan, const float *spec, const float *pos, int64_t n, int pos_mode,
                        int order, int lap_fd, int grad_fd, float *forces);
int mcpm_plan_force_meshes(mcpm_plan *plan, float **meshes3);

/* Kaiser-Bessel assignment kernel (nbody.py:280-290, selected by kernel_type = 'kaiser_bessel' in paint :381-382 and
   read :411-412): the same order^3 stencil, weights I0(kc sqrt(1 - (2 s / order)^2)) kc / (order sinh kc) per axis with
   kc = kcut order / 2 and kcut = optim_kcut(oversamp) (:357-363, computed by the caller).  order 1..4.
   mcpm_read_kb_f32: out (may be NULL) = read values; pos_bar (may be NULL) = out_bar_i d read_i / d pos (out_bar NULL:
   the scalar obscalar) -- with mesh = mesh_bar and out_bar = weights this is the position VJP of the paint, and `out`
   its weights VJP. */
int mcpm_paint_kb_f32(mcpm_plan *plan, const float *pos, int64_t n, int pos_mode, const float *weights, int64_t wstride,
                      float wscalar, int order, float kcut, float *mesh, int accumulate);
int mcpm_read_kb_f32(mcpm_plan *plan, const float *pos, int64_t n, int pos_mode, const float *mesh, int order, float kcut,
                     float *out, const float *out_bar, int64_t obstride, float obscalar, float *pos_bar);

/* ---- BullFrog / FastPM stepping (nbody.py:902-1002) ----------------------------------------- */
/* drift (nbody.py:942-944): pos_out = pos_in + vel * dt. */
int mcpm_drift_f32(mcpm_plan *plan, const float *pos_in, const float *vel, int64_t n, float dt, float *pos_out);
/* kick (nbody.py:933-938) given forces: vel_out = alpha * vel_in + beta * forces. */
int mcpm_kick_f32(mcpm_plan *plan, const float *vel_in, const float *forces, int64_t n, float alpha,
                  float beta, float *vel_out);
/* Fused CIC read of three force meshes + kick + drift: vel_out = alpha vel_in + beta F(pos_in),
   pos_out = pos_in + vel_out * dt. */
int mcpm_kick_drift_f32(mcpm_plan *plan, const float *pos_in, const float *vel_in, int64_t n, int pos_mode,
                        const float *meshes3, int order, float alpha, float beta, float dt, float *pos_out,
                        float *vel_out);
/* The same on ONE interleaved force mesh [x][y][z][3] (a CIC corner is one 12-byte gather for the three components; what
   mcpm_slab_zinv3_il writes). */
int mcpm_kick_drift_il_f32(mcpm_plan *plan, const float *pos_in, const float *vel_in, int64_t n, int pos_mode,
                           const float *mesh_il, int order, float alpha, float beta, float dt, float *pos_out,
                           float *vel_out);
/* slots: 64 x 32 device unsigned (zeroed by each kick_drift call) or NULL.  While set, every mcpm_kick_drift*_f32 leaves
   max_i |pos_out[i][0]| there as float bits (maximum over the 64 slots at stride 32): the slab stepper's ghost depth for
   the next step, without a pass of its own over the positions. */
int mcpm_plan_track_dmax(mcpm_plan *plan, unsigned *slots);
/* lpt (nbody.py:634-667) on the plan's particle lattice, read_order = 1, scalar a:
   dpos = g F1 - g2 F2, vel = F1 - dg2dg F2 (lpt_order 2) from the half-spectrum init_mesh. */
/* One BullFrog/FastPM step in the fused form x' -> paint -> forces -> v1 = alpha v + beta F(x'), x1 = x' + v1 tau
   (kick nbody.py:933-938 between the two half drifts nbody.py:946-950, adjacent half drifts merged).
   force_meshes (3 meshes, may be NULL -> plan scratch) receives the step's force meshes for the adjoint. */
int mcpm_bullfrog_step_f32(mcpm_plan *plan, const float *pos_in, const float *vel_in, double alpha, double beta,
                           double tau, int paint_order, float *force_meshes, float *pos_out, float *vel_out);
/* Adjoint of that step: pos_bar / vel_bar (cotangents of the step's outputs) are updated in place to the
   cotangents of its inputs; alpha_bar / beta_bar are DEVICE double accumulators (may be NULL).  dg_bar (DEVICE,
   may be NULL) accumulates the explicit dependence of the drift on the step size, <pos_bar_in, vel_out> * dtau_ddg
   with dtau_ddg = d tau / d dg (1, or 0.5 on the last step). */
int mcpm_bullfrog_step_vjp_f32(mcpm_plan *plan, const float *pos_in, const float *vel_in, const float *force_meshes,
                               double alpha, double beta, double tau, int paint_order, float *pos_bar,
                               float *vel_bar, double *alpha_bar, double *beta_bar, double dtau_ddg, double *dg_bar);
/* The same with the incoming cotangents read from (pos_bar_src, vel_bar_src) and the outgoing ones written to (pos_bar,
   vel_bar): the first reverse step of a trajectory can take the loss cotangents where they are instead of a copy of them (the
   cotangents of JAX's reverse sweep are fresh arrays at every step anyway: nbody.py:933-944 under jax.vjp).  src == out is the
   in-place form above. */
int mcpm_bullfrog_step_vjp_from_f32(mcpm_plan *plan, const float *pos_in, const float *vel_in, const float *force_meshes,
                                    double alpha, double beta, double tau, int paint_order, const float *pos_bar_src,
                                    const float *vel_bar_src, float *pos_bar, float *vel_bar, double *alpha_bar, double *beta_bar,
                                    double dtau_ddg, double *dg_bar);
/* Optional chaining of consecutive adjoint steps: call this before the adjoint of step i with beta and tau of step
   i-1; the particle kernel then also writes step i-1's force cotangent beta'(v_bar + tau' x_bar), and the next
   mcpm_bullfrog_step_vjp_f32 call skips its own pass over the cotangents IF it is given the same pos_bar / vel_bar
   pointers and exactly these scalars.  Only valid when the caller does not modify the cotangents in between. */
int mcpm_plan_hint_next_adjoint(mcpm_plan *plan, double beta_next, double tau_next);
/* After a hinted mcpm_step_adjoint_particles_f32: *fb = plan-owned F_bar = beta (v_bar + tau x_bar) (Np x 3 floats) of the
   next adjoint step if (beta, tau, pos_bar, vel_bar) are the hinted ones, else NULL (then the caller forms it itself with
   mcpm_kick_f32).  For callers that compose the adjoint step from the pieces (the slab-decomposed stepper). */
int mcpm_plan_chained_fb(mcpm_plan *plan, double beta, double tau, const float *pos_bar, const float *vel_bar, float **fb);
/* The particle half of that adjoint alone (fused gradient gather of the three force meshes and of rho_bar, kick /
   drift adjoints, scalar cotangents); used by the slab path, where the host exchanges ghost planes in between. */
int mcpm_step_adjoint_particles_f32(mcpm_plan *plan, const float *pos_in, const float *vel_in,
                                    const float *force_meshes, const float *rho_bar, double alpha, double beta,
                                    double tau, int paint_order, float *pos_bar, float *vel_bar, double *alpha_bar,
                                    double *beta_bar, double dtau_ddg, double *dg_bar);
/* The same with the step's force meshes as ONE interleaved mesh [x][y][z][3]. */
int mcpm_step_adjoint_particles_il_f32(mcpm_plan *plan, const float *pos_in, const float *vel_in,
                                       const float *force_mesh_il, const float *rho_bar, double alpha, double beta,
                                       double tau, int paint_order, float *pos_bar, float *vel_bar, double *alpha_bar,
                                       double *beta_bar, double dtau_ddg, double *dg_bar);
int mcpm_lpt_f32(mcpm_plan *plan, const float *init_mesh, int lpt_order, float g, float g2, float dg2dg,
                 int lap_fd, int grad_fd, float *dpos, float *vel);
/* Lattice-point pieces of lpt (read_order = 1 at pos = regular_pos, nbody.py:984-985), exposed for the slab path:
     lpt_accum      : dpos = (init ? 0 : dpos) + ad * F(q_i), vel likewise with av; F from three contiguous meshes
     lattice_scatter: its adjoint, meshes3[c][cell(q_i)] (+)= a * xb[i][c] + b * vb[i][c]
     lattice_dot    : out2[0] += sum_i a_i . F(q_i), out2[1] += sum_i b_i . F(q_i)   (DEVICE doubles; a or b may be NULL) */
int mcpm_lpt_accum_f32(mcpm_plan *plan, const float *meshes3, float ad, float av, int init, float *dpos, float *vel);
int mcpm_lattice_scatter_f32(mcpm_plan *plan, const float *xb, const float *vb, float a, float b, float *meshes3);
int mcpm_lattice_dot_f32(mcpm_plan *plan, const float *meshes3, const float *a, const float *b, double *out2);
/* VJP of mcpm_lpt_f32 w.r.t. init_mesh (real-pair convention, irfftn multiplicity weights included) and the three
   growth scalars: scalar_bars = {g_bar, g2_bar, dg2dg_bar} (host, may be NULL; forces a stream sync when given). */
int mcpm_lpt_vjp_f32(mcpm_plan *plan, const float *init_mesh, int lpt_order, const double *lpt_scalars,
                     const float *dpos_bar, const float *vel_bar, float *init_mesh_bar, double *scalar_bars);
/* The pair with the forward pass's meshes kept for the adjoint (ABI 0.6): `save` (caller-owned, 3 M floats for lpt_order 1, 12 M for 2)
   receives the first-order force meshes, the second-order ones and the six Hessian meshes; mcpm_lpt_vjp_saved_f32 reads them
   (`saved`) instead of recomputing them -- a third of the adjoint's transforms.  Infinite-order kernels only. */
int mcpm_lpt_save_f32(mcpm_plan *plan, const float *init_mesh, int lpt_order, float g, float g2, float dg2dg, float *dpos, float *vel,
                      float *save);
int mcpm_lpt_vjp_saved_f32(mcpm_plan *plan, const float *init_mesh, int lpt_order, const double *lpt_scalars, const float *saved,
                           const float *dpos_bar, const float *vel_bar, float *init_mesh_bar, double *scalar_bars);
/* The same with finite-difference kernels (lap_fd, grad_fd: MCPM_FD_*), as mcpm_lpt_f32 takes them. */
int mcpm_lpt_vjp_opts_f32(mcpm_plan *plan, const float *init_mesh, int lpt_order, const double *lpt_scalars, int lap_fd,
                          int grad_fd, const float *dpos_bar, const float *vel_bar, float *init_mesh_bar, double *scalar_bars);
/* nbody_bf (nbody.py:967-1002), snapshots=None: LPT start at a0 then n_steps drift-kick-drift steps of size
   dg in growth-factor time.  alpha[i] and beta[i] = (1-alpha_i)/(g_i + dg/2) are host float64 arrays computed
   from the growth tables (alpha_bf nbody.py:907-919 or alpha_fpm :921-931, evaluated at the accumulated Euler
   time g_i); `lpt_scalars` = {a2g, a2g2, a2dg2dg}(a0).  pos_out is the displacement from the lattice
   (MCPM_POS_LATTICE), vel_out the velocity.  ckpt (may be NULL) receives what the adjoint needs -- the step states, the steps' force
   meshes, and the force / Hessian meshes of the LPT start, which the reverse sweep then reads instead of recomputing them; its size is
   mcpm_nbody_ckpt_floats(plan, n_steps, lpt_order) floats. */
int mcpm_nbody_bf_f32(mcpm_plan *plan, const float *init_mesh, int n_steps, const double *alpha,
                      const double *beta, double dg, const double *lpt_scalars, int lpt_order, int paint_order,
                      float *pos_out, float *vel_out, float *ckpt);
int64_t mcpm_nbody_ckpt_floats(const mcpm_plan *plan, int n_steps, int lpt_order);
/* Layout of the particle arrays the composite entry points stream (the checkpoint states x'_i, v_i inside `ckpt`; the plan's own
   running cotangents): consecutive (N, 3) arrays are `pitch` floats apart, 3 N (back to back) by default.  The step kernels read
   and write four to seven such arrays at the SAME particle index; back to back they are in phase in every low address bit, and
   on some memory placements that a process draws they then meet in one memory channel (adjoint particle kernel 2.45 or 2.75 ms at
   512^3; DESIGN finding 29).  No shift is right everywhere, but a process can find out:
   mcpm_plan_probe_particle_pitch  times the adjoint particle kernel on the caller's checkpoint buffer `flat` (at least
       mcpm_nbody_ckpt_floats() floats; it is ZEROED -- call before filling it) for the pitches 3 N, 3 N + 1088 and 3 N + 17472
       floats, keeps the fastest for the following mcpm_nbody_bf_f32 / _vjp_f32 calls of this plan and returns it (meshes below
       2^23 particles, whose arrays live in the caches: 3 N, nothing is timed);
   mcpm_plan_set_particle_pitch    fixes it (0 = back to back; else 3 N <= pitch <= 3 N + 17472, a multiple of 4);
   mcpm_plan_particle_pitch        the pitch in force: state (x'_i, v_i) of a checkpoint = arrays 2 i and 2 i + 1 of `ckpt`.
   Callers of the step-level entry points (mcpm_bullfrog_step_f32 ...) own their arrays: the same advice applies to them
   (INTEGRATION.md).  No reference counterpart (XLA places the reference's buffers). */
int mcpm_plan_probe_particle_pitch(mcpm_plan *plan, float *flat, int64_t flat_floats, int64_t *pitch_floats);
int mcpm_plan_set_particle_pitch(mcpm_plan *plan, int64_t pitch_floats);
int mcpm_plan_particle_pitch(const mcpm_plan *plan, int64_t *pitch_floats);
/* Reverse sweep: cotangents of (pos_out, vel_out) -> init_mesh_bar (half-spectrum, real-pair convention
   dL = Re sum conj(bar) dz) and host scalar bars (may be NULL; forces a stream sync when given):
   scalar_bars[0..n_steps) = alpha_bar, [n_steps..2 n_steps) = beta_bar, then {g_bar, g2_bar, dg2dg_bar, dg_bar}
   (2 n_steps + 4 doubles; dg_bar is the explicit dependence of the drifts on the step size). */
int mcpm_nbody_bf_vjp_f32(mcpm_plan *plan, const float *init_mesh, int n_steps, const double *alpha,
                          const double *beta, double dg, const double *lpt_scalars, int lpt_order,
                          int paint_order, const float *ckpt, const float *pos_bar, const float *vel_bar,
                          float *init_mesh_bar, double *scalar_bars);

/* ---- host-side float64 growth tables (nbody.py:679-745) ------------------------------------- */
/* rg2cgh / cgh2rg with norm = "backward" (montecosmo/utils.py:785-921): a real Gaussian tensor (nx, ny, nz), all sizes
   even, <-> the complex Hermitian tensor (nx, ny, nz/2+1; plain complex64) distributed as rfftn of a real Gaussian tensor
   (signed permutation with sqrt(M/2) / sqrt(M) weights).  The VJP of rg2cgh takes the cotangent of the complex tensor
   (real-pair convention) to that of the real one.  No plan: `stream` is a hipStream_t. */
int mcpm_rg2cgh_f32(void *stream, const float *real, int nx, int ny, int nz, float *spec);
int mcpm_rg2cgh_vjp_f32(void *stream, const float *spec_bar, int nx, int ny, int nz, float *real_bar);
int mcpm_cgh2rg_f32(void *stream, const float *spec, int nx, int ny, int nz, float *real);
/* cgh2rg with norm = "amp" (utils.py:916-918): the real AND the imaginary element of every mode take Re spec there,
   unsigned and unweighted -- a per-mode amplitude laid out like the real tensor (model.py:1147, the 'kaiser' prior scale). */
int mcpm_cgh2rg_amp_f32(void *stream, const float *spec, int nx, int ny, int nz, float *real);

/* Lagrangian bias expansion (montecosmo/bricks.py:327-443, png_type = None).
   fields : lin_mesh (plain half-spectrum of the plan's mesh) -> fields7 = {delta, shear^2, 3 det(shear), laplacian(delta),
            grad_x, grad_y, grad_z} as 7 real meshes M apart; wavevectors in h/Mpc, kphys[a] = mesh_shape[a] / box_size[a]
            (bricks.py:352).  The VJP maps the 7 cotangent meshes to the cotangent of lin_mesh (real-pair convention).
   weights: the raw reads of those fields at the particles (dr, s2r, s3r, lr: n floats each; gr: n x 3) -> weights (n) and
            dvel (n x 3).  gr_cstride = 0: gr (and gr_bar) are particle-major (n x 3); > 0: component-major, component c at
            gr + c * gr_cstride (the three gradient meshes themselves when the particles are the mesh's own lattice and the
            read is NGP: then no read / paint pass is needed at all).  growth = a2g(a): one float per particle, or NULL
            and growth_scalar.  bias8 (host) =
            {b1, b2, bs2, b3, bds2, bs3, bn2, bnpar}.  The VJP returns the cotangents of the raw reads, of growth (per
            particle if growth_bar != NULL) and scalars_out (device, 10 doubles) = 8 bias cotangents, summed growth
            cotangent, <d^2>.  The reads between the two halves are mcpm_read_f32; their adjoints mcpm_paint_f32. */
int mcpm_bias_fields_f32(mcpm_plan *plan, const float *lin_mesh, float kphys_x, float kphys_y, float kphys_z, float *fields7);
int mcpm_bias_fields_vjp_f32(mcpm_plan *plan, const float *lin_mesh, float kphys_x, float kphys_y, float kphys_z,
                             const float *fields7_bar, float *lin_mesh_bar);
/* The same pair with the forward pass's intermediates kept for the adjoint: hess6 (6 M floats, caller-owned; NULL = not kept) receives
   delta and the five Hessian meshes of the shear; mcpm_bias_fields_vjp_saved_f32 reads them instead of recomputing them from
   lin_mesh (six transforms less per gradient; 400 MB at 256^3). */
int mcpm_bias_fields_save_f32(mcpm_plan *plan, const float *lin_mesh, float kphys_x, float kphys_y, float kphys_z, float *fields7,
                              float *hess6);
int mcpm_bias_fields_vjp_saved_f32(mcpm_plan *plan, float kphys_x, float kphys_y, float kphys_z, const float *hess6,
                                   const float *fields7_bar, float *lin_mesh_bar);
int mcpm_bias_weights_f32(mcpm_plan *plan, int64_t n, const float *dr, const float *s2r, const float *s3r, const float *lr,
                          const float *gr, int64_t gr_cstride, const float *growth, float growth_scalar, const float *bias8,
                          float *weights, float *dvel, double *sigma2_out);
int mcpm_bias_weights_vjp_f32(mcpm_plan *plan, int64_t n, const float *dr, const float *s2r, const float *s3r, const float *lr,
                              const float *gr, int64_t gr_cstride, const float *growth, float growth_scalar, const float *bias8,
                              const float *weights_bar, const float *dvel_bar, float *dr_bar, float *s2r_bar, float *s3r_bar,
                              float *lr_bar, float *gr_bar, float *growth_bar, double *scalars_out);

/* white2lin / lin2white multiplier (montecosmo/bricks.py:83-100, :149-161): out = in * sqrt(amp * P(|k|)), |k| in h/Mpc
   (kphys = mesh_shape / box_size), P linearly interpolated from the DEVICE float64 table (ks ascending, pows) and zero
   outside it (jnp.interp left = right = 0); amp = sigma8^2 for a table normalised to sigma8 = 1.  Real multiplier: the same
   call is its own adjoint. */
int mcpm_power_mult_f32(mcpm_plan *plan, const float *in, float kphys_x, float kphys_y, float kphys_z, double amp,
                        const double *ks, const double *pows, int ntab, float *out);
/* out[i] = scale * np.interp(x[i], xp, fp) (clamped ends; the growth / distance look-ups of montecosmo/nbody.py:748-804,
   :862-884 for per-particle scale factors on the light cone).  Tables float64 on the DEVICE, xp ascending. */
int mcpm_interp_f32(mcpm_plan *plan, const float *x, int64_t n, const double *xp, const double *fp, int ntab, float scale,
                    float *out);
/* Light-cone LPT (montecosmo/nbody.py:652-666 with a of shape (N,1)): per-particle growth (gtab (n,3) = a2g, a2g2, a2dg2dg at
   a_i) applied to the first / second order forces at the particles: dpos = g F1 - g2 F2, vel = F1 - dg2dg F2 (F2 NULL for
   lpt_order 1).  F2 and F1 are what mcpm_lpt_f32 returns as (dpos, vel) for (g, g2, dg2dg) = (0, -1, 0).  The VJP works in
   place: (xb, vb) cotangents of (dpos, vel) become those of (F2, F1), ready for mcpm_lpt_vjp_f32 with the same scalars. */
int mcpm_lpt_combine_f32(mcpm_plan *plan, const float *F1, const float *F2, const float *gtab, int64_t n, float *dpos, float *vel);
int mcpm_lpt_combine_vjp_f32(mcpm_plan *plan, const float *F1, const float *F2, const float *gtab, int64_t n, float *xb, float *vb,
                             float *gtab_bar);

/* Evolved particles -> redshift-space positions on the paint mesh (montecosmo/model.py:780-797 without Alcock-Paczynski;
   bricks.py:628-662 cell <-> physical maps, :750-768 line of sight and scale factor, :791-803 rsd), one fused pass.
   The plan's mesh is the evolution mesh.  geom (host, 19 floats) = box_rot matrix R[9] (row major, apply(x) = R x),
   box_size[3], box_center[3], paint_shape[3], g(a_obs) f(a_obs).  flags: bit 0 = curved sky, bit 1 = light cone; on the
   light cone `tables` (DEVICE float64) = chi ascending [nchi], a(chi) [nchi], a [ngrow], g [ngrow], f [ngrow].
   pos_mode MCPM_POS_LATTICE: pos = displacements from the plan's particle lattice, out = displacements from that lattice
   scaled to the paint mesh; MCPM_POS_ABSOLUTE: absolute cell coordinates in and out.  dvel (Mpc/h, may be NULL) is the
   bias velocity term.  The VJP returns the cotangents of pos, vel, dvel and of the scalar g f (device double; 0 on the
   light cone, where the cosmology dependence of the tables is not propagated). */
int mcpm_observe_pos_f32(mcpm_plan *plan, const float *pos, const float *vel, const float *dvel, int64_t n, int pos_mode,
                         const float *geom, int flags, const double *tables, int nchi, int ngrow, float *out);
int mcpm_observe_pos_vjp_f32(mcpm_plan *plan, const float *pos, const float *vel, const float *dvel, int64_t n, int pos_mode,
                             const float *geom, int flags, const double *tables, int nchi, int ngrow, const float *out_bar,
                             float *pos_bar, float *vel_bar, float *dvel_bar, double *gf_bar);
/* Light cone (a_obs = None, the reference's default: model.py:62): how the cosmology enters the per-particle look-ups.
   The scale factor of a particle is chi2a(distance) and its growth quantities are table look-ups at that scale factor
   (model.py:740-742 -> bricks.py:750-768 los_scalefactor_pos -> nbody.py:862-884 chi2a, :748-808 a2g ...; again at the evolved
   positions for the RSD, model.py:781-784).  These two calls contract per-particle cotangents into the cotangents of the TABLES
   themselves (float64 on the device; nodes of the chi -> a look-up and values of the growth tables), which the host chains to
   cosmological parameters with the tables' finite-difference Jacobian (montecosmo_amd/model.py cosmo_vjp).  The sums are taken
   in 64-bit fixed point (a first pass finds the scale), so they are bitwise the same call after call.
   mcpm_lightcone_tables_vjp_f32: Lagrangian side.  r0 (n): particle distances; tables = chi[nchi] ascending, a(chi)[nchi],
   a[ngrow], g, g2 (raw table, without the -3/7), f, f2 [ngrow each]; g_bar / g2_bar / dg2dg_bar (n floats; the last two may be
   NULL): cotangents of a2g(a), a2g2(a), a2dg2dg(a).  table_bar (OVERWRITTEN): chi_bar[nchi], g_bar, g2_bar, f_bar, f2_bar [ngrow each].
   mcpm_observe_pos_tables_vjp_f32: observation side, arguments of mcpm_observe_pos_vjp_f32 (flags must carry the light-cone bit);
   table_bar (OVERWRITTEN): chi_bar[nchi], g_bar[ngrow], f_bar[ngrow]. */
int mcpm_lightcone_tables_vjp_f32(mcpm_plan *plan, const float *r0, int64_t n, const double *tables, int nchi, int ngrow,
                                  const float *g_bar, const float *g2_bar, const float *dg2dg_bar, double *table_bar);
int mcpm_observe_pos_tables_vjp_f32(mcpm_plan *plan, const float *pos, const float *vel, const float *dvel, int64_t n, int pos_mode,
                                    const float *geom, int flags, const double *tables, int nchi, int ngrow, const float *out_bar,
                                    double *table_bar);

/* chreshape (montecosmo/utils.py:924-1013): half-spectrum of a real (in_nx, in_ny, in_nz) mesh -> half-spectrum of a
   real (out_nx, out_ny, out_nz) mesh, truncating / zero-padding the centred wavevectors with the reference's Nyquist-plane
   aggregation (1/sqrt2 weights) and cell-count scale, so Hermitian symmetry and mean power are preserved.  Plain
   complex64 layout (nx, ny, nz/2+1) on both sides, all sizes even; no plan: `stream` is a hipStream_t.  The VJP takes the
   cotangent of the output (real-pair convention dL = Re sum conj(bar) dz) and writes the cotangent of the input. */
int mcpm_chreshape_c64(void *stream, const float *in, int in_nx, int in_ny, int in_nz, float *out, int out_nx, int out_ny,
                       int out_nz);
int mcpm_chreshape_vjp_c64(void *stream, const float *out_bar, int out_nx, int out_ny, int out_nz, float *in_bar, int in_nx,
                           int in_ny, int in_nz);

/* RK4 on atab = logspace(log10_amin, 0, steps); writes seven host arrays of length `steps`. */
int mcpm_growth_table(double Omega_m, double Omega_de, double Omega_k, double w0, double wa,
                      double log10_amin, int steps, double *a, double *g, double *f, double *h, double *g2,
                      double *f2, double *h2);
/* chi(a) table of nbody.py:842-856 (256-point RK4 in ln a); host arrays of length `steps`. */
int mcpm_distance_table(double Omega_m, double Omega_de, double Omega_k, double w0, double wa,
                        double log10_amin, int steps, double *a, double *chi);

#ifdef __cplusplus
}
#endif
#endif /* MCPM_H */
