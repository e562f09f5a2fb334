// Plan lifetime, error plumbing and the rocFFT R2C / C2R wrappers of libmcpm.so.
// Replaces the jnp.fft.rfftn / irfftn calls of montecosmo/nbody.py:589, :603, :620, :627, :630.
#include <mutex>

#include "mcpm_internal.h"

thread_local std::string g_mcpm_create_error;

int mcpm_fail(mcpm_plan *plan, int code, const std::string &msg) {
    if (plan)
        plan->err = msg;
    else
        g_mcpm_create_error = msg;
    return code;
}

static std::once_flag g_rocfft_once;

static int make_fft(mcpm_plan *p, bool forward, int batch) {
    auto &plans = forward ? p->r2c : p->c2r;
    if (plans.count(batch)) return MCPM_OK;
    size_t lengths[3] = {(size_t)p->g.nz, (size_t)p->g.ny, (size_t)p->g.nx};
    rocfft_plan plan = nullptr;
    rocfft_status st = rocfft_plan_create(&plan, rocfft_placement_notinplace,
                                          forward ? rocfft_transform_type_real_forward : rocfft_transform_type_real_inverse,
                                          rocfft_precision_single, 3, lengths, (size_t)batch, nullptr);
    if (st != rocfft_status_success) return mcpm_fail(p, MCPM_E_ROCFFT, "rocfft_plan_create failed: " + std::to_string((int)st));
    size_t wsz = 0;
    rocfft_plan_get_work_buffer_size(plan, &wsz);
    rocfft_execution_info info = nullptr;
    rocfft_execution_info_create(&info);
    void *work = nullptr;
    if (wsz) {
        if (hipMalloc(&work, wsz) != hipSuccess) {
            rocfft_execution_info_destroy(info);
            rocfft_plan_destroy(plan);
            return mcpm_fail(p, MCPM_E_NOMEM, "rocFFT work buffer");
        }
        rocfft_execution_info_set_work_buffer(info, work, wsz);
    }
    rocfft_execution_info_set_stream(info, p->stream);
    plans[batch] = plan;
    (forward ? p->r2c_info : p->c2r_info)[batch] = info;
    (forward ? p->r2c_work : p->c2r_work)[batch] = work;
    return MCPM_OK;
}

extern "C" {

const char *mcpm_version(void) { return MCPM_ABI_VERSION; }

const char *mcpm_last_error(const mcpm_plan *plan) { return plan ? plan->err.c_str() : g_mcpm_create_error.c_str(); }

static int plan_create_impl(int nx, int ny, int nz, int px, int py, int pz, int nranks, int rank, int ghost, void *stream,
                            mcpm_plan **out);

int mcpm_plan_create(int nx, int ny, int nz, int px, int py, int pz, void *stream, mcpm_plan **out) {
    return plan_create_impl(nx, ny, nz, px, py, pz, 1, 0, 0, stream, out);
}

int mcpm_plan_create_slab(int nx, int ny, int nz, int nranks, int rank, int ghost, void *stream, mcpm_plan **out) {
    if (!out) return mcpm_fail(nullptr, MCPM_E_ARG, "plan output pointer is null");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return mcpm_fail(nullptr, MCPM_E_ARG, "bad rank / nranks");
    if (nx % nranks || ny % nranks) return mcpm_fail(nullptr, MCPM_E_SHAPE, "nx and ny must be divisible by the number of ranks");
    const int nxl = nx / nranks;
    if (ghost < 5 || ghost > nxl) return mcpm_fail(nullptr, MCPM_E_ARG, "ghost width must be in [5, nx/nranks]");
    return plan_create_impl(nx, ny, nz, nxl, ny, nz, nranks, rank, ghost, stream, out);
}

// Halo H of the tiled paints' windows ((16 + 2H + 1)^3 lattice points per tile).  What it should be depends on how much the
// displacements vary within a tile's neighbourhood: every cell of halo costs visits ((25/23)^3 = +28 % from 3 to 4), every
// particle the windows miss costs the exact coverage test and a bucket deposit.  On the benchmark family (rms displacement 2
// cells; the smaller the mesh, the rougher the field per cell) 4 wins up to 256^3 and 3 above: 64^3 6725 vs 6314 steps/s, 128^3
// 3635 vs 3231, 192^3 993 vs 864, 256^3 689 vs 703 (but pm_forces on the evolved particles 0.552 vs 0.630 ms), 512^3 87.8 vs 90.0.
// The choice never changes a result beyond the last bit (the sums are exact; the split between tile and bucket deposits moves).
// This static rule is what slab plans and MCPM_PAINT_ADAPT=0 use; periodic plans choose H per input on the device
// (paint_tiled.hip::box_tile_kernel).  MCPM_PAINT_HALO / mcpm_plan_set_halo fix it.
// (mcpm_default_halo: mcpm_internal.h)

// nranks == 1, ghost == 0: ordinary periodic plan.  Otherwise (slab): (nx, ny, nz) is the GLOBAL mesh, the local
// mesh is the ghost-extended slab (nx/nranks + 2 ghost, ny, nz) and the lattice (px, py, pz) = (nx/nranks, ny, nz).
static int plan_create_impl(int nx, int ny, int nz, int px, int py, int pz, int nranks, int rank, int ghost, void *stream,
                            mcpm_plan **out) {
    if (!out) return mcpm_fail(nullptr, MCPM_E_ARG, "plan output pointer is null");
    *out = nullptr;
    const bool slab = ghost > 0;
    const int nx_global = nx;
    if (slab) nx = px + 2 * ghost;
    if (nx < 2 || ny < 2 || nz < 2 || (nz & 1)) return mcpm_fail(nullptr, MCPM_E_SHAPE, "mesh dims must be >= 2 and nz even");
    if (nx > 32766 || ny > 32766 || nz > 32766) return mcpm_fail(nullptr, MCPM_E_SHAPE, "mesh side must stay below 32767 (int16 index math)");
    if (px < 1 || py < 1 || pz < 1) return mcpm_fail(nullptr, MCPM_E_SHAPE, "particle lattice dims must be >= 1");
    if ((int64_t)px * py * pz >= (int64_t)1 << 31) return mcpm_fail(nullptr, MCPM_E_SHAPE, "more than 2^31 particles per plan");
    mcpm_plan *p = new (std::nothrow) mcpm_plan();
    if (!p) return mcpm_fail(nullptr, MCPM_E_NOMEM, "host allocation");
    p->g = Geom{nx, ny, nz, px, py, pz, nz / 2 + 1, ((slab || px == nx) && py == ny && pz == nz) ? 1 : 0, slab ? ghost : 0, slab ? 1 : 0, 0};
    {   // see mcpm_plan_set_lattice_patch
        const char *e = getenv("MCPM_LATTICE_PATCH");
        p->g.patch = (px % 2 == 0 && py % 2 == 0 && pz % 64 == 0 && !(e && atoi(e) == 0)) ? 1 : 0;
    }
    p->nranks = nranks;
    p->rank = rank;
    p->ghost = ghost;
    p->nx_global = nx_global;
    p->nxl = slab ? px : nx;
    p->dmax = nullptr;
    p->xw0 = 0;
    p->xwn = p->nxl;
    p->chunks = 1;
    p->slab_state = nullptr;
    p->stream = (hipStream_t)stream;
    p->M = (int64_t)nx * ny * nz;
    p->Mh = (int64_t)nx * ny * p->g.nzh;
    p->Np = (int64_t)px * py * pz;
    p->halo = 0;      // 0: chosen per input on the device (paint_tiled.hip::halo_of / box_tile_kernel), else mcpm_default_halo
    p->centre = 1;    // windows centred on the bulk displacement; on the tile itself (centre = 0) they need one more cell of halo at the
                      // benchmark's 2-cell rms displacement (bench 512^3: 12.10 vs 12.42 ms per step, pm_forces 4.38 vs 4.48 ms)
    if (const char *e = getenv("MCPM_PAINT_CENTRE")) p->centre = atoi(e) ? 1 : 0;
    if (const char *e = getenv("MCPM_PAINT_HALO")) { const int h = atoi(e); if (h == 1 || h == 2 || h == 3 || h == 4 || h == 6) p->halo = h; }
    p->tile_off = p->bucket_cnt = p->bucket = p->bucket_tiles = p->halo_sel = nullptr;
    p->bucket_cap = 0;
    p->paint_variant = 0;
    if (const char *e = getenv("MCPM_PAINT_VARIANT")) p->paint_variant = atoi(e);
    p->paint3_variant = 4;   // fixed-point tiles (particles.hip); 2 = f64 tiles
    p->hint_set = 0;
    p->fb_valid = 0;
    if (const char *e = getenv("MCPM_PAINT3_VARIANT")) p->paint3_variant = atoi(e);
    p->rho = p->spec = p->fmesh = p->spec1 = p->fft_pad = nullptr;
    p->outliers = p->outlier_count = nullptr;
    p->fx_wmax = nullptr;
    p->fx_redo = nullptr;
    p->fx_tiles = 0;
    p->fx_src = nullptr;
    p->fx_clean = 0;
    p->gx_acc = nullptr;
    p->gx_wmax = nullptr;
    p->reduce = nullptr;
    p->pscratch = nullptr;
    p->ppitch = 0;
    p->part = nullptr;
    p->part_n = 0;
    p->vscratch = nullptr;
    p->vscratch_n = 0;
    p->tw[0] = p->tw[1] = p->tw[2] = nullptr;
    p->profiling = 0;
    std::call_once(g_rocfft_once, [] { rocfft_setup(); });
    hipError_t e = hipSuccess;
    auto alloc = [&](void **ptr, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(ptr, bytes);
    };
    alloc((void **)&p->rho, sizeof(float) * p->M);
    {   // 6 plain half-spectra (rocFFT path) or 1 + 6 spectra in the padded layout of the hand-written FFT
        const size_t plain = (size_t)p->Mh * 6, padded = (size_t)nx * ny * (nz / 2 + MCPM_NZPAD) * 7;
        alloc((void **)&p->spec, sizeof(float) * 2 * (plain > padded ? plain : padded));
    }
    alloc((void **)&p->fmesh, sizeof(float) * p->M * 9);
    alloc((void **)&p->spec1, sizeof(float) * 2 * p->Mh);
    alloc((void **)&p->outliers, sizeof(int) * 2 * p->Np);   // suspects | wild particles
    alloc((void **)&p->outlier_count, sizeof(int) * 8);
    if (p->g.same_lattice && nx % 16 == 0 && ny % 16 == 0 && nz % 16 == 0 && nx >= 48 && ny >= 48 && nz >= 48) {
        const size_t ntiles = (size_t)(nx / 16) * (ny / 16) * (nz / 16);
        p->bucket_cap = 1024;
        if (const char *e = getenv("MCPM_BUCKET_CAP")) { const int c = atoi(e); if (c >= 64 && c <= 65536) p->bucket_cap = c; }
        alloc((void **)&p->tile_off, sizeof(int) * ntiles);
        alloc((void **)&p->halo_sel, sizeof(int) * 3 * ntiles);      // sampled floor(d) ranges of the Lagrangian blocks + the windows' upper corners (paint_tiled.hip)
        alloc((void **)&p->bucket_cnt, sizeof(int) * ntiles);
        alloc((void **)&p->bucket_tiles, sizeof(int) * ntiles);
        alloc((void **)&p->bucket, sizeof(int) * ntiles * p->bucket_cap);
    }
    p->fx_tiles = (int)(p->M / 4096 + 2);
    alloc((void **)&p->fx_wmax, sizeof(unsigned) * MCPM_FX_SLOTS * MCPM_FX_STRIDE);
    alloc((void **)&p->fx_redo, sizeof(int) * (p->fx_tiles + 1));
    alloc((void **)&p->gx_wmax, sizeof(unsigned) * MCPM_FX_SLOTS * MCPM_FX_STRIDE);
    alloc((void **)&p->reduce, sizeof(double) * MCPM_NREDUCE);
    if (e != hipSuccess) {
        std::string msg = std::string("hipMalloc of plan scratch: ") + hipGetErrorString(e);
        mcpm_plan_destroy(p);
        return mcpm_fail(nullptr, MCPM_E_NOMEM, msg);
    }
    (void)hipMemsetAsync(p->outlier_count, 0, sizeof(int) * 8, p->stream);
    (void)hipMemsetAsync(p->fx_redo, 0, sizeof(int), p->stream);
    (void)hipMemsetAsync(p->reduce, 0, sizeof(double) * MCPM_NREDUCE, p->stream);   // the slot area stays zero between uses
    *out = p;
    return MCPM_OK;
}

int mcpm_plan_destroy(mcpm_plan *p) {
    if (!p) return MCPM_OK;
    (void)hipStreamSynchronize(p->stream);
    for (auto &r : p->recs) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    for (auto &e : p->event_pool) (void)hipEventDestroy(e);
    for (auto &kv : p->r2c) rocfft_plan_destroy(kv.second);
    for (auto &kv : p->c2r) rocfft_plan_destroy(kv.second);
    for (auto &kv : p->r2c_info) rocfft_execution_info_destroy(kv.second);
    for (auto &kv : p->c2r_info) rocfft_execution_info_destroy(kv.second);
    for (auto &kv : p->r2c_work) (void)hipFree(kv.second);
    for (auto &kv : p->c2r_work) (void)hipFree(kv.second);
    (void)hipFree(p->rho);
    (void)hipFree(p->spec);
    (void)hipFree(p->fmesh);
    (void)hipFree(p->spec1);
    (void)hipFree(p->fft_pad);
    mcpm_slab_state_free(p);
    (void)hipFree(p->outliers);
    (void)hipFree(p->outlier_count);
    (void)hipFree(p->tile_off);
    (void)hipFree(p->halo_sel);
    (void)hipFree(p->bucket_cnt);
    (void)hipFree(p->bucket_tiles);
    (void)hipFree(p->bucket);
    (void)hipFree(p->fx_wmax);
    (void)hipFree(p->fx_redo);
    (void)hipFree(p->gx_acc);
    (void)hipFree(p->gx_wmax);
    (void)hipFree(p->reduce);
    (void)hipFree(p->pscratch);
    (void)hipFree(p->part);
    (void)hipFree(p->vscratch);
    for (int a = 0; a < 3; ++a) (void)hipFree(p->tw[a]);
    delete p;
    return MCPM_OK;
}

int mcpm_plan_set_halo(mcpm_plan *p, int halo) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, halo == 0 || halo == 1 || halo == 2 || halo == 3 || halo == 4 || halo == 6, MCPM_E_ARG, "halo must be 0 (default), 1, 2, 3, 4 or 6");
    p->halo = halo;   // 0: per input, on the device
    return MCPM_OK;
}

int mcpm_plan_set_lattice_patch(mcpm_plan *p, int on) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, !on || (p->g.px % 2 == 0 && p->g.py % 2 == 0 && p->g.pz % 64 == 0), MCPM_E_SHAPE,
                 "mcpm_plan_set_lattice_patch: needs an even px, py and pz a multiple of 64");
    p->g.patch = on ? 1 : 0;
    return MCPM_OK;
}

int mcpm_plan_set_centre(mcpm_plan *p, int centre) {
    if (!p) return MCPM_E_ARG;
    p->centre = centre ? 1 : 0;
    return MCPM_OK;
}

}  // extern "C"

int mcpm_det_scratch(mcpm_plan *p, int K, unsigned nblk, double **P, double **Q, unsigned **ticket, unsigned *R) {
    // Few first-stage workgroups: each ends with one atomic on the shared ticket, and those serialise at ~27 ns apiece (1024 of them:
    // 34 us per fold at 512^3, against 23 us with 256); 128 workgroups still pull 12.6 MB of partials in a few microseconds.
    const unsigned r = std::max(1u, std::min(128u, (nblk + 511u) / 512u));
    // [ticket (one double slot, kept zero)] [Q: K * 1024] [P: K * nblk]
    const int64_t need = 1 + (int64_t)K * 1024 + (int64_t)K * nblk;
    if (p->part_n < need) {
        if (p->part) {
            MCPM_HIP(p, hipStreamSynchronize(p->stream));
            (void)hipFree(p->part);
            p->part = nullptr;
        }
        const int64_t n = need + need / 4;
        if (hipMalloc((void **)&p->part, sizeof(double) * n) != hipSuccess) return mcpm_fail(p, MCPM_E_NOMEM, "reduction scratch");
        p->part_n = n;
        MCPM_HIP(p, hipMemsetAsync(p->part, 0, sizeof(double), p->stream));
    }
    *ticket = reinterpret_cast<unsigned *>(p->part);
    *Q = p->part + 1;
    *P = p->part + 1 + (int64_t)K * 1024;
    *R = r;
    return MCPM_OK;
}

extern "C" {

int mcpm_plan_last_bucketed(mcpm_plan *p, int64_t *count) {
    if (!p || !count) return MCPM_E_ARG;
    int h = 0;
    MCPM_HIP(p, hipMemcpyAsync(&h, p->outlier_count + 5, sizeof(int), hipMemcpyDeviceToHost, p->stream));
    MCPM_HIP(p, hipStreamSynchronize(p->stream));
    *count = h;
    return MCPM_OK;
}

// out[0..6]: the last tiled paint's device counters (wild particles, last wild + overflow pairs, slab deposits beyond the ghost planes,
// appends that found their bucket full, tiles with a non-empty bucket, bucketed pairs, suspects); out[7]: window points of all tiles
// (/ particles = window visits per particle); out[8 + h], h = 0..4: tiles whose widest window axis has 17 + 2h - 1 or 17 + 2h points
// (what a symmetric halo of h would give; all zero if the plan's windows are not centred).  Synchronises the host.
int mcpm_plan_last_paint_stats(mcpm_plan *p, int64_t *out13) {
    if (!p || !out13) return MCPM_E_ARG;
    int c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    MCPM_HIP(p, hipMemcpyAsync(c, p->outlier_count, sizeof(c), hipMemcpyDeviceToHost, p->stream));
    for (int i = 0; i < 13; ++i) out13[i] = 0;
    std::vector<int> lo, hi;
    if (p->tile_off && p->halo_sel && p->centre) {
        const size_t nt = (size_t)(p->M / 4096);
        lo.resize(nt), hi.resize(nt);
        MCPM_HIP(p, hipMemcpyAsync(lo.data(), p->tile_off, sizeof(int) * nt, hipMemcpyDeviceToHost, p->stream));
        MCPM_HIP(p, hipMemcpyAsync(hi.data(), p->halo_sel + 2 * nt, sizeof(int) * nt, hipMemcpyDeviceToHost, p->stream));
    }
    MCPM_HIP(p, hipStreamSynchronize(p->stream));
    for (int i = 0; i < 7; ++i) out13[i] = c[i];
    for (size_t t = 0; t < lo.size(); ++t) {
        int64_t pts = 1;
        int ext = 0;
        for (int a = 0; a < 3; ++a) {
            const int e = (int)(int8_t)((hi[t] >> (8 * a)) & 0xff) - (int)(int8_t)((lo[t] >> (8 * a)) & 0xff);
            pts *= 17 + e;
            ext = std::max(ext, e);
        }
        out13[7] += pts;
        out13[8 + std::min((ext + 1) / 2, 4)] += 1;
    }
    return MCPM_OK;
}

int mcpm_plan_set_paint3_fixed(mcpm_plan *p, int fixed) {
    if (!p) return MCPM_E_ARG;
    p->paint3_variant = fixed ? 4 : 2;
    p->fx_src = nullptr;
    return MCPM_OK;
}

int mcpm_plan_last_redo(mcpm_plan *p, int64_t *count) {
    if (!p || !count) return MCPM_E_ARG;
    int h = 0;
    MCPM_HIP(p, hipMemcpyAsync(&h, p->fx_redo, sizeof(int), hipMemcpyDeviceToHost, p->stream));
    MCPM_HIP(p, hipStreamSynchronize(p->stream));
    *count = h;
    return MCPM_OK;
}

int mcpm_plan_last_outliers(mcpm_plan *p, int64_t *count) {
    if (!p || !count) return MCPM_E_ARG;
    int h = 0;
    MCPM_HIP(p, hipMemcpyAsync(&h, p->outlier_count + 1, sizeof(int), hipMemcpyDeviceToHost, p->stream));
    MCPM_HIP(p, hipStreamSynchronize(p->stream));
    *count = h;
    return MCPM_OK;
}

static const char *k_stage_names[ST_NSTAGES] = {"paint", "fft_r2c", "fft_c2r", "kspace", "read", "kick_drift",
                                                 "step_adjoint", "axpy", "lpt_lattice", "paint3"};

const char *mcpm_stage_name(int stage) { return (stage >= 0 && stage < ST_NSTAGES) ? k_stage_names[stage] : ""; }

int mcpm_plan_profile(mcpm_plan *p, int enable) {
    if (!p) return MCPM_E_ARG;
    p->profiling = enable ? 1 : 0;
    return MCPM_OK;
}

int mcpm_plan_profile_read(mcpm_plan *p, int nmax, double *ms, double *bytes, int64_t *calls) {
    if (!p || !ms || !bytes || !calls) return MCPM_E_ARG;
    MCPM_HIP(p, hipStreamSynchronize(p->stream));
    for (int i = 0; i < nmax; ++i) {
        ms[i] = 0.;
        bytes[i] = 0.;
        calls[i] = 0;
    }
    for (auto &r : p->recs) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.e0, r.e1);
        if (r.stage < nmax) {
            ms[r.stage] += t;
            bytes[r.stage] += r.bytes;
            calls[r.stage] += 1;
        }
        p->event_pool.push_back(r.e0);
        p->event_pool.push_back(r.e1);
    }
    p->recs.clear();
    return ST_NSTAGES;
}

int mcpm_plan_slab_oob(mcpm_plan *p, int64_t *count) {
    if (!p || !count) return MCPM_E_ARG;
    int h = 0;
    MCPM_HIP(p, hipMemcpyAsync(&h, p->outlier_count + 2, sizeof(int), hipMemcpyDeviceToHost, p->stream));
    MCPM_HIP(p, hipStreamSynchronize(p->stream));
    *count = h;
    return MCPM_OK;
}

int mcpm_plan_hint_next_adjoint(mcpm_plan *p, double beta_next, double tau_next) {
    if (!p) return MCPM_E_ARG;
    p->hint_beta = (float)beta_next;
    p->hint_tau = (float)tau_next;
    p->hint_set = 1;
    return MCPM_OK;
}

int mcpm_plan_force_meshes(mcpm_plan *p, float **meshes3) {
    if (!p || !meshes3) return MCPM_E_ARG;
    *meshes3 = p->fmesh;
    return MCPM_OK;
}

int mcpm_fft_r2c(mcpm_plan *p, const float *real, float *spec, int batch) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, real && spec && batch >= 1, MCPM_E_ARG, "mcpm_fft_r2c: null buffer or batch < 1");
    if (!p->g.xslab && mcpm_fftpm_supported(p) && !getenv("MCPM_FORCE_ROCFFT")) return mcpm_fftpm_r2c(p, real, spec, batch);
    MCPM_TRY(make_fft(p, true, batch));
    StageTimer st_(p, ST_R2C, (double)batch * (4.0 * p->M + 8.0 * p->Mh));
    void *in[1] = {(void *)real};
    void *out[1] = {(void *)spec};
    rocfft_status st = rocfft_execute(p->r2c[batch], in, out, p->r2c_info[batch]);
    if (st != rocfft_status_success) return mcpm_fail(p, MCPM_E_ROCFFT, "rocfft_execute r2c: " + std::to_string((int)st));
    return MCPM_OK;
}

int mcpm_fft_c2r(mcpm_plan *p, float *spec, float *real, int batch) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, real && spec && batch >= 1, MCPM_E_ARG, "mcpm_fft_c2r: null buffer or batch < 1");
    if (!p->g.xslab && mcpm_fftpm_supported(p) && !getenv("MCPM_FORCE_ROCFFT")) return mcpm_fftpm_c2r(p, spec, real, batch);
    MCPM_TRY(make_fft(p, false, batch));
    StageTimer st_(p, ST_C2R, (double)batch * (4.0 * p->M + 8.0 * p->Mh));
    void *in[1] = {(void *)spec};
    void *out[1] = {(void *)real};
    rocfft_status st = rocfft_execute(p->c2r[batch], in, out, p->c2r_info[batch]);
    if (st != rocfft_status_success) return mcpm_fail(p, MCPM_E_ROCFFT, "rocfft_execute c2r: " + std::to_string((int)st));
    return MCPM_OK;
}

}  // extern "C"
