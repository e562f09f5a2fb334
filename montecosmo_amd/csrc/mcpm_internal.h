// Internal declarations shared by the HIP translation units of libmcpm.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/mcpm.h"

// The store-data hazard of wide vector-memory stores.  The rule (LLVM AMDGPU GCNHazardRecognizer::checkVALUHazardsHelper /
// createsVALUHazard, "VMEM instructions that store more than 8 bytes can have their store data overwritten by the next
// instruction"): a FLAT / global / buffer store whose data operand is wider than 64 bits reads its data VGPRs after issue, and a
// VALU instruction that WRITES one of them must be at least VALUWaitStates behind the store -- 1 wait state up to gfx90a, **2 on
// parts with the gfx940 instruction set (ST.hasGFX940Insts(), which gfx950 has)**.  The compiler pads its own stores; it does
// not look inside inline asm, so every hand-written global_store_dwordx3 below carries the padding itself: `s_nop 2` = 3 wait
// states (s_nop N idles N + 1), one more than the rule asks.  Without it a few cells in ten thousand received whatever the
// next instructions had put into the data registers (round 3, gpurun_out/r03l).  mcpm_selftest_store3_nt (particles.hip) runs
// this exact sequence with the data registers overwritten by the very next instructions; tests/test_gpu_parity.py holds it to
// a plain-store twin over 2^26 records.
#define MCPM_STORE_DATA_HAZARD_NOP "s_nop 2"


#define MCPM_NREDUCE 6144   // [0, 3072): scalar outputs and the bias / observe slot rows; [3072, 6144): the 3 x MCPM_NSLOT step-adjoint slots (kept zero between uses)
#ifndef MCPM_NZPAD
#define MCPM_NZPAD 16  // complex elements added to the nz/2 pitch of the internal spectra
#endif

// Stages of the path, for the optional per-stage HIP-event profile (mcpm_plan_profile*).
enum McpmStage {
    ST_PAINT = 0,   // paint_tile_kernel + paint_outlier_kernel, or paint_atomic_kernel
    ST_R2C,         // rocFFT real forward
    ST_C2R,         // rocFFT real inverse
    ST_KSPACE,      // k-space force / hessian kernels and their adjoints
    ST_READ,        // read / read_vjp / paint_vjp gathers
    ST_KICKDRIFT,   // fused read + kick + drift
    ST_STEPADJ,     // fused adjoint particle kernel
    ST_AXPY,        // drift / kick / cotangent axpy
    ST_LPT,         // lattice kernels of lpt and its adjoint, hessian combine
    ST_PAINT3,      // three-component tiled paint (adjoint of the three-component read)
    ST_NSTAGES
};

struct StageRec {
    hipEvent_t e0, e1;
    int stage;
    double bytes;
};

// Mesh + particle-lattice geometry handed to kernels by value.
struct Geom {
    int nx, ny, nz;   // mesh
    int px, py, pz;   // particle lattice (regular_pos(mesh, ptcl))
    int nzh;          // nz/2 + 1
    int same_lattice; // px==nx && py==ny && pz==nz (unit lattice spacing)
    int xoff;         // slab mode: mesh plane of lattice plane 0 (= ghost width); 0 otherwise
    int xslab;        // slab mode: x is NOT periodic on this (ghost-extended) mesh
    int patch;        // lattice-mode particle kernels: a 256-thread workgroup is four waves on a 2 x 2 (x, y) patch of lattice rows
                      // (64 consecutive z each) instead of 256 consecutive z of one row (particles_dev.h::particle_index)
};

#define MCPM_FX_SLOTS 64
#define MCPM_FX_STRIDE 32   // unsigned per slot: one 128-byte line each

struct mcpm_plan {
    Geom g;
    hipStream_t stream;
    int64_t M;   // nx*ny*nz
    int64_t Mh;  // nx*ny*nzh
    int64_t Np;  // px*py*pz
    int halo;    // tiled paints: > 0 = fixed halo H (a tile's window is (16 + 2H + 1)^3 lattice points around its bulk offset); 0 = boxes chosen on the device
    int centre;  // tiled paints: windows centred on the local bulk displacement (paint_tiled.hip); 0 = on the tile itself
    int *halo_sel;    // 3 x ntiles packed words: per 16^3 Lagrangian block the sampled range of floor(d) per axis ([ntiles] minima, [ntiles] maxima), from which
                      // box_tile_kernel sizes every tile's window, then [ntiles] upper corners of the windows
    int *tile_off;    // packed lower corners of the windows' floor(d) boxes per 16^3 tile (device; NULL if the mesh cannot be tiled)
    int *bucket_cnt;  // per-tile bucket fill counts
    int *bucket;      // [tile][bucket_cap] particles a tile's window misses
    int bucket_cap;
    int *bucket_tiles; // list of tiles with a non-empty bucket (built by the duty, walked by the bucket kernels)
    int paint_variant;  // threads/unroll variant of the tiled paint (tuning)
    int paint3_variant; // three-component tiled paint variant (tuning); < 0 disables it; 4 = fixed-point tile
    unsigned *fx_wmax;  // fixed-point paint: bits of max|w|, maximum over MCPM_FX_SLOTS slots (device)
    int *fx_redo;       // fixed-point paint: [0] = number of flagged tiles, then their indices (device)
    int fx_tiles;       // capacity of fx_redo
    const float *fx_src; // weights whose max|w| a producer kernel already left in fx_wmax (else NULL)
    int fx_clean;        // fx_wmax is all zero (the last three-component paint's epilogue cleared it): producers skip their memset
    long long *gx_acc;  // generic (order-independent) paint: int64 fixed-point accumulator mesh, all-zero between calls; allocated on first use
    unsigned *gx_wmax;  // generic paint: bits of max|w| (MCPM_FX_SLOTS slots)
    // chaining of adjoint steps (mcpm_plan_hint_next_adjoint): the adjoint particle kernel of step i also writes the
    // force cotangent F_bar of step i-1, saving one pass over the cotangents
    int hint_set, fb_valid;
    float hint_beta, hint_tau, fb_beta, fb_tau;
    const void *fb_xb, *fb_vb;
    // x-slab decomposition (mcpm_plan_create_slab): this rank owns global planes [rank*nxl, (rank+1)*nxl)
    int nranks, rank, ghost, nx_global, nxl;
    unsigned *dmax; // caller's MCPM_FX_SLOTS x MCPM_FX_STRIDE slots: kick_drift leaves max |d_x| (as float bits) there; NULL = off
    int xw0, xwn;  // window of local planes the slab z / y passes work on (mcpm_slab_set_window; default all nxl)
    int chunks;    // chunks of the all-to-all layouts (mcpm_slab_set_chunks; 1 = one all-to-all per spectrum)
    void *slab_state;  // transport + workspace of the native slab steps (slab.hip); NULL until mcpm_slab_comm_init_*

    // rocFFT plans keyed by batch
    std::map<int, rocfft_plan> r2c, c2r;
    std::map<int, rocfft_execution_info> r2c_info, c2r_info;
    std::map<int, void *> r2c_work, c2r_work;

    // scratch owned by the plan
    float *rho;      // M floats: painted density / real scratch
    float *spec;     // 6 half-spectra (complex64) scratch
    float *fmesh;    // 9 real meshes scratch (force meshes / hessians)
    float *spec1;    // 1 half-spectrum scratch
    float *fft_pad;  // 1 padded spectrum: scratch of the generic hand-written R2C / C2R (allocated on first use)
    int *outliers;   // lists of the tiled paint: suspects (Np ints), then wild particles (Np ints)
    int *outlier_count;  // device counters (8 ints, see paint_tiled.hip: wild, last, slab oob, overflow pairs, dropped, bucketed)
    double *reduce;  // device accumulators for scalar cotangents (MCPM_NREDUCE doubles)
    float *pscratch; // 3 (N, 3) arrays mcpm_pitch_max() apart, allocated on first VJP (adjoint state x_bar, v_bar + force cotangent F_bar)
    // Pitch (floats) between consecutive (N, 3) particle arrays of the composite entry points' checkpoint (mcpm_nbody_bf_f32: x'_0, v_0,
    // x'_1, ...) and of pscratch; 0 = contiguous (3 N).  The step kernels stream four to seven such arrays at the same particle index:
    // back to back they are IN PHASE in every low address bit and may meet in one memory channel, depending on where the process's
    // memory was placed (DESIGN finding 29); a pitch of 3 N + a few KB takes them out of phase.  mcpm_plan_probe_particle_pitch
    // measures the candidates on the caller's buffer; mcpm_plan_set_particle_pitch fixes one.
    int64_t ppitch;
    float *vscratch;  // variable-size particle scratch (pm_forces_vjp)
    double *part;     // per-workgroup partial sums of the deterministic grid reductions (reduce_dev.h), allocated on first use
    int64_t part_n;
    int64_t vscratch_n;
    float *tw[3];    // twiddle tables exp(-2 pi i j / n) of the hand-written FFT, per axis (x, y, z)

    // optional profile: HIP events recorded on the plan's stream around every leaf stage
    int profiling;
    std::vector<StageRec> recs;
    std::vector<hipEvent_t> event_pool;

    std::string err;
};

// RAII stage bracket: records two events on the plan's stream when profiling is on.
struct StageTimer {
    mcpm_plan *p;
    StageRec r;
    bool on;
    StageTimer(mcpm_plan *plan, int stage, double bytes) : p(plan), on(plan && plan->profiling) {
        if (!on) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!p->event_pool.empty()) {
                e = p->event_pool.back();
                p->event_pool.pop_back();
            } else {
                (void)hipEventCreate(&e);
            }
            return e;
        };
        r.e0 = get();
        r.e1 = get();
        r.stage = stage;
        r.bytes = bytes;
        (void)hipEventRecord(r.e0, p->stream);
    }
    ~StageTimer() {
        if (!on) return;
        (void)hipEventRecord(r.e1, p->stream);
        p->recs.push_back(r);
    }
};

extern thread_local std::string g_mcpm_create_error;

int mcpm_fail(mcpm_plan *plan, int code, const std::string &msg);
// Scratch of one deterministic grid reduction (reduce_dev.h): K values from nblk workgroups.  *P: K * nblk partials, *Q: K * R second-level
// sums, *ticket: zero between launches; *R: workgroups of det_fold_kernel.  (plan.hip)
int mcpm_det_scratch(mcpm_plan *p, int K, unsigned nblk, double **P, double **Q, unsigned **ticket, unsigned *R);
void mcpm_slab_state_free(mcpm_plan *p);   // slab.hip
#define MCPM_PITCH_MAX_SHIFT 17472      // floats: 64 KB + 4 KB + 256 B, the largest candidate shift between particle arrays
static inline int64_t mcpm_pitch_max(const mcpm_plan *p) { return 3 * p->Np + MCPM_PITCH_MAX_SHIFT; }
static inline int64_t mcpm_pitch(const mcpm_plan *p) { return p->ppitch > 0 ? p->ppitch : 3 * p->Np; }
static inline int mcpm_default_halo(int64_t M) { return M <= ((int64_t)1 << 24) ? 4 : 3; }   // the static rule (plan.hip: where it comes from)

// hand-written FFT Poisson solve (fftpm.hip); power-of-two axes only
bool mcpm_fftpm_supported(const mcpm_plan *p);
int mcpm_fftpm_force_meshes(mcpm_plan *p, const float *rho, float *fm3, int interleaved = 0, int nt_out = 0);
int mcpm_kick_drift_layout(mcpm_plan *p, const float *pos_in, const float *vel_in, int64_t n, int mode, const float *meshes3,
                                      int layout, int order, float alpha, float beta, float dt, float *pos_out, float *vel_out);
int mcpm_read3_il(mcpm_plan *p, const float *pos, int64_t n, int mode, const float *fm_il, int order, float *out);
int mcpm_fftpm_force_meshes_vjp(mcpm_plan *p, const float *fbar3, float *rho_bar);
int mcpm_fftpm_r2c(mcpm_plan *p, const float *real, float *spec, int batch);
int mcpm_fftpm_c2r(mcpm_plan *p, const float *spec, float *real, int batch);
// plain half-spectrum -> nc = 3 force meshes or nc = 6 Hessian meshes (00 01 02 11 12 22), and the adjoints
// (spec_bar overwritten for nc = 3, accumulated into for nc = 6)
int mcpm_fftpm_spec_meshes(mcpm_plan *p, const float *spec, float *meshes, int nc);
int mcpm_fftpm_spec_meshes_vjp(mcpm_plan *p, const float *meshes_bar, float *spec_bar, int nc);

// tiled CIC paints (paint_tiled.hip); false: geometry not tileable, the caller takes the generic path
bool mcpm_paint_tiled(mcpm_plan *p, const float *pos, const float *w, int64_t wstride, float wscalar, float *mesh, int accumulate);
bool mcpm_paint3_tiled(mcpm_plan *p, const float *pos, const float *weights3, float *meshes3, int accumulate);

// adjoint of the NGP lattice read on a lattice != mesh: order-independent fixed-point sums (particles.hip)
int mcpm_lattice_scatter_fx(mcpm_plan *p, const float *xb, const float *vb, float a, float b, float *meshes3);

#define MCPM_HIP(plan, expr)                                                                         \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return mcpm_fail(plan, MCPM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

#define MCPM_LAUNCH_CHECK(plan, name)                                                                \
    do {                                                                                             \
        hipError_t _e = hipGetLastError();                                                           \
        if (_e != hipSuccess)                                                                        \
            return mcpm_fail(plan, MCPM_E_HIP, std::string("launch ") + name + ": " + hipGetErrorString(_e)); \
    } while (0)

#define MCPM_REQUIRE(plan, cond, code, msg)                  \
    do {                                                     \
        if (!(cond)) return mcpm_fail(plan, code, msg);      \
    } while (0)

#define MCPM_TRY(expr)               \
    do {                             \
        int _rc = (expr);            \
        if (_rc != MCPM_OK) return _rc; \
    } while (0)
