// One slab-decomposed BullFrog step and its adjoint behind the C ABI: the library issues the FFT transposes (all-to-all) and
// the ghost-plane exchanges (point-to-point with the two x neighbours) itself, on its own communication stream, ordered
// against the plan's compute stream with events -- one host call per step instead of ~36 kernel-level calls and ~10
// torch.distributed collectives (montecosmo_amd/dist.py SlabPM.step_gen / step_vjp_gen, which stays as the tested
// reference: this file issues the same kernels on the same windows in the same order, so results are bitwise equal).
//
// No counterpart in the reference (its only multi-device mode is independent chains, montecosmo/script.py:13-20); the
// contract is SURVEY.md 8(b) "the library owns ... the RCCL communicator" and 8(e).
//
// Transports (mcpm_slab_comm_init_*):
//   local : one rank; ghost exchanges are device copies of the own planes, a one-rank transpose is the identity (aliased);
//   rccl  : librccl is opened at run time (dlopen: the copy the process already holds, else the ROCm one), one communicator
//           per plan, every exchange is ONE ncclGroup of ncclSend / ncclRecv on the plan's communication stream;
//   ops   : callbacks supplied by the host (tests: several gloo ranks sharing one GPU drive the multi-rank index algebra of
//           this file through the host).
#include "mcpm_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>

namespace {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllToAll) AllToAll = nullptr;     // RCCL extension (optional): the equal-split transposes
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
};

RcclApi *rccl_api() {
    static RcclApi api = [] {
        RcclApi a;
        // the copy already mapped into the process first (torch ships its own librccl.so): two RCCLs in one process would each
        // keep their own device state
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if ((a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!a.lib)
            for (const char *n : names)
                if ((a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!a.lib) {
            a.why = std::string("librccl not found: ") + dlerror();
            return a;
        }
#define SYM(name)                                                             \
    a.name = (decltype(a.name))dlsym(a.lib, "nccl" #name);                    \
    if (!a.name) {                                                            \
        a.why = "librccl lacks nccl" #name;                                   \
        a.lib = nullptr;                                                      \
        return a;                                                             \
    }
        SYM(GetUniqueId) SYM(CommInitRank) SYM(CommDestroy) SYM(Send) SYM(Recv) SYM(GroupStart) SYM(GroupEnd) SYM(AllReduce)
        SYM(GetErrorString)
#undef SYM
        a.AllToAll = (decltype(a.AllToAll))dlsym(a.lib, "ncclAllToAll");
        return a;
    }();
    return &api;
}

enum { MODE_NONE = 0, MODE_LOCAL = 1, MODE_RCCL = 2, MODE_OPS = 3 };
constexpr int NTICKETS = 96;   // exchanges of one step never exceed this many in flight

struct SlabState {
    int mode = MODE_NONE;
    ncclComm_t comm = nullptr;
    hipStream_t cs = nullptr;                  // communication stream (rccl)
    hipEvent_t ready[NTICKETS], done[NTICKETS];
    bool events = false;
    int next_ticket = 0;
    mcpm_comm_ops ops{};
    // workspace bound by the caller (mcpm_slab_bind_workspace)
    float *rho = nullptr, *f3 = nullptr, *s1a = nullptr, *s1b = nullptr, *s6a = nullptr, *s6b = nullptr, *Fb = nullptr, *halo = nullptr;
    // ghost-depth measurement: max |d_x| over ranks of the last kick_drift, on its way to pinned host memory
    // (a ring: the host reads a measurement one step late, when its device work has long finished, and never stops for it)
    float *dmax_dev = nullptr, *dmax_host = nullptr;      // dmax_host: DMAX_RING pinned floats
    hipEvent_t dmax_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t dmax_seq = 0;                                 // measurements enqueued so far
};
constexpr int DMAX_RING = 4;

SlabState *state(mcpm_plan *p) { return static_cast<SlabState *>(p->slab_state); }

int rccl_fail(mcpm_plan *p, ncclResult_t r, const char *what) {
    RcclApi *a = rccl_api();
    return mcpm_fail(p, MCPM_E_RCCL, std::string(what) + ": " + (a->GetErrorString ? a->GetErrorString(r) : "RCCL error"));
}
#define RCCL_TRY(p, call)                                          \
    do {                                                           \
        ncclResult_t r_ = (call);                                  \
        if (r_ != ncclSuccess) return rccl_fail((p), r_, #call);   \
    } while (0)

int ensure_events(mcpm_plan *p, SlabState *s) {
    if (s->events) return MCPM_OK;
    for (int i = 0; i < NTICKETS; ++i) {
        MCPM_HIP(p, hipEventCreateWithFlags(&s->ready[i], hipEventDisableTiming));
        MCPM_HIP(p, hipEventCreateWithFlags(&s->done[i], hipEventDisableTiming));
    }
    s->events = true;
    return MCPM_OK;
}

// ---- one batch of point-to-point transfers -------------------------------------------------------------------
struct Batch {
    std::vector<const void *> sp;
    std::vector<void *> rp;
    std::vector<int64_t> sb, rb;
    std::vector<int> speer, rpeer;
    // an equal-split all-to-all (block q of `a2a_in` to rank q, block q of `a2a_out` from rank q): transports with a native
    // all-to-all take it whole instead of the sends / receives above, which describe the same transfer
    const void *a2a_in = nullptr;
    void *a2a_out = nullptr;
    int64_t a2a_bytes = 0;
    void send(const void *ptr, int64_t bytes, int peer) { sp.push_back(ptr); sb.push_back(bytes); speer.push_back(peer); }
    void recv(void *ptr, int64_t bytes, int peer) { rp.push_back(ptr); rb.push_back(bytes); rpeer.push_back(peer); }
};

// Starts the batch, ordered after everything already enqueued on the compute stream; *ticket < 0: already complete.
int xfer_begin(mcpm_plan *p, const Batch &b, int *ticket) {
    SlabState *s = state(p);
    *ticket = -1;
    if (s->mode == MODE_LOCAL) {   // every peer is this rank: the k-th send meets the k-th receive
        if (b.sp.size() != b.rp.size()) return mcpm_fail(p, MCPM_E_ARG, "local exchange: sends and receives do not pair");
        for (size_t i = 0; i < b.sp.size(); ++i) {
            if (b.sb[i] != b.rb[i]) return mcpm_fail(p, MCPM_E_ARG, "local exchange: size mismatch");
            if (b.sp[i] != b.rp[i]) MCPM_HIP(p, hipMemcpyAsync(b.rp[i], b.sp[i], (size_t)b.sb[i], hipMemcpyDeviceToDevice, p->stream));
        }
        return MCPM_OK;
    }
    if (s->mode == MODE_OPS) {
        int t = -1;
        const int rc = s->ops.p2p_begin(s->ops.ctx, (int)b.sp.size(), b.sp.data(), b.sb.data(), b.speer.data(), (int)b.rp.size(),
                                        b.rp.data(), b.rb.data(), b.rpeer.data(), (void *)p->stream, &t);
        if (rc != 0) return mcpm_fail(p, MCPM_E_RCCL, "transport callback p2p_begin failed");
        *ticket = t;
        return MCPM_OK;
    }
    if (s->mode != MODE_RCCL) return mcpm_fail(p, MCPM_E_ARG, "no transport: call mcpm_slab_comm_init_* first");
    RcclApi *a = rccl_api();
    const int t = s->next_ticket;
    s->next_ticket = (t + 1) % NTICKETS;
    MCPM_HIP(p, hipEventRecord(s->ready[t], p->stream));
    MCPM_HIP(p, hipStreamWaitEvent(s->cs, s->ready[t], 0));
    if (b.a2a_in && a->AllToAll) {
        RCCL_TRY(p, a->AllToAll(b.a2a_in, b.a2a_out, (size_t)b.a2a_bytes, ncclInt8, s->comm, s->cs));
    } else {
        // a failing Send / Recv must not leave the group open (every later call on this thread would join it): remember the
        // first error, close the group whatever happened, report afterwards
        RCCL_TRY(p, a->GroupStart());
        ncclResult_t first = ncclSuccess;
        const char *what = "";
        for (size_t i = 0; i < b.sp.size() && first == ncclSuccess; ++i) {
            first = a->Send(b.sp[i], (size_t)b.sb[i], ncclInt8, b.speer[i], s->comm, s->cs);
            what = "ncclSend";
        }
        for (size_t i = 0; i < b.rp.size() && first == ncclSuccess; ++i) {
            first = a->Recv(b.rp[i], (size_t)b.rb[i], ncclInt8, b.rpeer[i], s->comm, s->cs);
            what = "ncclRecv";
        }
        const ncclResult_t end = a->GroupEnd();
        if (first != ncclSuccess) return rccl_fail(p, first, what);
        if (end != ncclSuccess) return rccl_fail(p, end, "ncclGroupEnd");
    }
    MCPM_HIP(p, hipEventRecord(s->done[t], s->cs));
    *ticket = t;
    return MCPM_OK;
}

// Orders the compute stream after the batch.
int xfer_wait(mcpm_plan *p, int ticket) {
    if (ticket < 0) return MCPM_OK;
    SlabState *s = state(p);
    if (s->mode == MODE_OPS) {
        if (s->ops.wait(s->ops.ctx, ticket, (void *)p->stream) != 0) return mcpm_fail(p, MCPM_E_RCCL, "transport callback wait failed");
        return MCPM_OK;
    }
    MCPM_HIP(p, hipStreamWaitEvent(p->stream, s->done[ticket], 0));
    return MCPM_OK;
}

// ---- geometry helpers (mirror dist.SlabPM) --------------------------------------------------------------------
struct Win {
    int x0, n;
};
struct Part {
    bool has_inner;
    Win inner;
    std::vector<Win> edges;
};

// Per chunk: interior planes and the edge plane runs a pending ghost exchange touches (SlabPM._chunk_parts)
std::vector<Part> chunk_parts(const mcpm_plan *p, int d, bool pending) {
    const int nxl = p->nxl, C = p->chunks > 0 ? p->chunks : 1, cw = nxl / C;
    std::vector<Part> out;
    for (int w = 0; w < C; ++w) {
        const int lo = w * cw, hi = (w + 1) * cw;
        Part pt;
        if (pending && nxl <= 2 * d) {
            pt.has_inner = false;
            pt.edges.push_back(Win{lo, hi - lo});
            out.push_back(pt);
            continue;
        }
        const int a = pending ? std::max(lo, d) : lo, b = pending ? std::min(hi, nxl - d) : hi;
        if (a > lo) pt.edges.push_back(Win{lo, a - lo});
        if (hi > b) pt.edges.push_back(Win{b, hi - b});
        pt.has_inner = b > a;
        pt.inner = Win{a, b - a};
        out.push_back(pt);
    }
    return out;
}

std::vector<int> edge_first(int C) {   // SlabPM._edge_first
    std::vector<int> o{0};
    if (C > 1) o.push_back(C - 1);
    for (int w = 1; w < C - 1; ++w) o.push_back(w);
    return o;
}

struct Ctx {
    mcpm_plan *p;
    SlabState *s;
    int64_t plane, Me, ss;   // floats per mesh plane, floats per ghost-extended mesh, complex per spectrum
    int G, nxl, P, r, C, d;
    float *interior(float *ext) const { return ext + (int64_t)G * plane; }
    float *interior_il(float *ext) const { return ext + 3 * (int64_t)G * plane; }
    float *spec(float *buf, int c) const { return buf + 2 * (int64_t)c * ss; }
    int win(Win w) const { return mcpm_slab_set_window(p, w.x0, w.n); }
    int win_all() const { return mcpm_slab_set_window(p, 0, nxl); }
};

// all-to-all of chunk w of one spectrum (ss complex at `in` -> `out`); *aliased: the result stayed in `in`
int a2a_chunk(const Ctx &c, float *out, const float *in, int w, int *ticket, bool *aliased) {
    *ticket = -1;
    if (c.s->mode == MODE_LOCAL) {
        *aliased = true;
        return MCPM_OK;
    }
    *aliased = false;
    const int64_t n = c.ss / c.C, blk = n / c.P;     // complex per chunk region / per peer
    Batch b;
    for (int q = 0; q < c.P; ++q) b.send(in + 2 * (w * n + q * blk), 8 * blk, q);
    for (int q = 0; q < c.P; ++q) b.recv(out + 2 * (w * n + q * blk), 8 * blk, q);
    b.a2a_in = in + 2 * w * n;
    b.a2a_out = out + 2 * w * n;
    b.a2a_bytes = 8 * blk;
    return xfer_begin(c.p, b, ticket);
}

// ghost exchange of `nc` arrays that are `cstride` floats apart, each with x leading and `pf` floats per plane
// fill: my first / last d interior planes go INTO the neighbours' ghost planes (in place)
int halo_fill_begin(const Ctx &c, float *ext, int nc, int64_t cstride, int64_t pf, int *ticket) {
    const int left = (c.r - 1 + c.P) % c.P, right = (c.r + 1) % c.P, G = c.G, d = c.d, nxl = c.nxl;
    const int64_t run = 4 * (int64_t)d * pf;
    Batch b;
    for (int k = 0; k < nc; ++k) b.send(ext + k * cstride + (int64_t)G * pf, run, left);                       // -> left's high ghost
    for (int k = 0; k < nc; ++k) b.send(ext + k * cstride + (int64_t)(G + nxl - d) * pf, run, right);         // -> right's low ghost
    for (int k = 0; k < nc; ++k) b.recv(ext + k * cstride + (int64_t)(G + nxl) * pf, run, right);             // my high ghost
    for (int k = 0; k < nc; ++k) b.recv(ext + k * cstride + (int64_t)(G - d) * pf, run, left);                // my low ghost
    return xfer_begin(c.p, b, ticket);
}

// add: my ghost planes are ADDED into the neighbours' interiors; receives land in the halo scratch (2 nc runs of G planes)
int halo_add_begin(const Ctx &c, float *ext, int nc, int64_t cstride, int *ticket) {
    const int left = (c.r - 1 + c.P) % c.P, right = (c.r + 1) % c.P, G = c.G, d = c.d, nxl = c.nxl;
    const int64_t pf = c.plane, run = 4 * (int64_t)d * pf, slot = (int64_t)G * pf;
    float *from_l = c.s->halo, *from_r = c.s->halo + nc * slot;
    Batch b;
    for (int k = 0; k < nc; ++k) b.send(ext + k * cstride + (int64_t)(G - d) * pf, run, left);                 // my low ghost -> left
    for (int k = 0; k < nc; ++k) b.send(ext + k * cstride + (int64_t)(G + nxl) * pf, run, right);             // my high ghost -> right
    for (int k = 0; k < nc; ++k) b.recv(from_r + k * slot, run, right);
    for (int k = 0; k < nc; ++k) b.recv(from_l + k * slot, run, left);
    return xfer_begin(c.p, b, ticket);
}

int halo_add_finish(const Ctx &c, float *ext, int nc, int64_t cstride, int ticket) {
    MCPM_TRY(xfer_wait(c.p, ticket));
    const int G = c.G, d = c.d, nxl = c.nxl;
    const int64_t pf = c.plane, slot = (int64_t)G * pf, n = (int64_t)d * pf;
    float *from_l = c.s->halo, *from_r = c.s->halo + nc * slot;
    for (int k = 0; k < nc; ++k) {
        float *lo = ext + k * cstride + (int64_t)G * pf, *hi = ext + k * cstride + (int64_t)(G + nxl - d) * pf;
        MCPM_TRY(mcpm_axpby_f32(c.p, lo, from_l + k * slot, n, 1.f, 1.f, lo));    // left neighbour's high ghost = my lowest planes
        MCPM_TRY(mcpm_axpby_f32(c.p, hi, from_r + k * slot, n, 1.f, 1.f, hi));    // right neighbour's low ghost = my highest planes
    }
    return MCPM_OK;
}

int make_ctx(mcpm_plan *p, int depth, Ctx *c) {
    SlabState *s = state(p);
    MCPM_REQUIRE(p, s && s->mode != MODE_NONE, MCPM_E_ARG, "slab step: no transport (mcpm_slab_comm_init_*)");
    MCPM_REQUIRE(p, s->rho && s->f3 && s->s1a && s->s1b && s->s6a && s->s6b && s->Fb && s->halo, MCPM_E_ARG,
                 "slab step: no workspace (mcpm_slab_bind_workspace)");
    MCPM_REQUIRE(p, p->ghost > 0, MCPM_E_ARG, "slab step needs a slab plan (mcpm_plan_create_slab)");
    MCPM_REQUIRE(p, depth >= 1 && depth <= p->ghost, MCPM_E_ARG, "slab step: ghost depth outside [1, ghost]");
    c->p = p;
    c->s = s;
    c->plane = (int64_t)p->g.ny * p->g.nz;
    c->Me = (int64_t)p->g.nx * c->plane;
    c->ss = mcpm_slab_spec_elems(p);
    c->G = p->ghost;
    c->nxl = p->nxl;
    c->P = p->nranks;
    c->r = p->rank;
    c->C = p->chunks > 0 ? p->chunks : 1;
    c->d = depth;
    return MCPM_OK;
}

__global__ void dmax_reduce_kernel(const unsigned *__restrict__ slots, float *__restrict__ out) {
    unsigned m = slots[(threadIdx.x & (MCPM_FX_SLOTS - 1)) * MCPM_FX_STRIDE];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if (threadIdx.x == 0) out[0] = __uint_as_float(m);      // float bits of non-negative values order like unsigned
}

// max |d_x| over ranks of the positions kick_drift just wrote -> pinned host memory, without stopping the host
int dmax_enqueue(const Ctx &c) {
    mcpm_plan *p = c.p;
    SlabState *s = c.s;
    if (!p->dmax) return MCPM_OK;
    if (!s->dmax_dev) {
        MCPM_HIP(p, hipMalloc((void **)&s->dmax_dev, sizeof(float)));
        MCPM_HIP(p, hipHostMalloc((void **)&s->dmax_host, DMAX_RING * sizeof(float), hipHostMallocDefault));
        for (int i = 0; i < DMAX_RING; ++i) MCPM_HIP(p, hipEventCreateWithFlags(&s->dmax_ev[i], hipEventDisableTiming));
    }
    dmax_reduce_kernel<<<1, 64, 0, p->stream>>>(p->dmax, s->dmax_dev);
    MCPM_LAUNCH_CHECK(p, "dmax_reduce_kernel");
    if (s->mode == MODE_RCCL && c.P > 1) {
        RcclApi *a = rccl_api();
        const int t = s->next_ticket;
        s->next_ticket = (t + 1) % NTICKETS;
        MCPM_HIP(p, hipEventRecord(s->ready[t], p->stream));
        MCPM_HIP(p, hipStreamWaitEvent(s->cs, s->ready[t], 0));
        RCCL_TRY(p, a->AllReduce(s->dmax_dev, s->dmax_dev, 1, ncclFloat32, ncclMax, s->comm, s->cs));
        MCPM_HIP(p, hipEventRecord(s->done[t], s->cs));
        MCPM_HIP(p, hipStreamWaitEvent(p->stream, s->done[t], 0));
    } else if (s->mode == MODE_OPS) {
        if (s->ops.allreduce_max_f32(s->ops.ctx, s->dmax_dev, (void *)p->stream) != 0)
            return mcpm_fail(p, MCPM_E_RCCL, "transport callback allreduce_max_f32 failed");
    }
    const int slot = (int)(s->dmax_seq % DMAX_RING);
    MCPM_HIP(p, hipMemcpyAsync(s->dmax_host + slot, s->dmax_dev, sizeof(float), hipMemcpyDeviceToHost, p->stream));
    MCPM_HIP(p, hipEventRecord(s->dmax_ev[slot], p->stream));
    s->dmax_seq += 1;
    return MCPM_OK;
}

// rho (interior of the ghost-extended slab; ghost add `add_ticket` possibly in flight) -> interleaved force mesh with the
// ghost planes filled (SlabPM.force_meshes_gen, il = True)
int force_meshes(const Ctx &c, float *rho_ext, float *f3_il, bool add_pending, int add_ticket) {
    mcpm_plan *p = c.p;
    SlabState *s = c.s;
    const int C = c.C;
    float *s3a = s->s6a, *s3b = s->s6b;
    std::vector<Part> parts = chunk_parts(p, c.d, add_pending);
    std::vector<int> h1(C, -1);
    bool aliased = false;
    auto zy = [&](Win w) -> int {
        MCPM_TRY(c.win(w));
        MCPM_TRY(mcpm_slab_zfwd(p, c.interior(rho_ext), c.Me, s->s1a, 1));
        return mcpm_slab_ycol(p, s->s1a, s->s1b, 1, -1, 0, 1);       // plain -> transposed order
    };
    for (int w = 0; w < C; ++w) {
        if (parts[w].has_inner) MCPM_TRY(zy(parts[w].inner));
        if (parts[w].edges.empty()) MCPM_TRY(a2a_chunk(c, s->s1a, s->s1b, w, &h1[w], &aliased));
    }
    if (add_pending) MCPM_TRY(halo_add_finish(c, rho_ext, 1, 0, add_ticket));
    for (int w = 0; w < C; ++w) {
        for (const Win &e : parts[w].edges) MCPM_TRY(zy(e));
        if (!parts[w].edges.empty()) MCPM_TRY(a2a_chunk(c, s->s1a, s->s1b, w, &h1[w], &aliased));
    }
    MCPM_TRY(c.win_all());
    float *x_in = aliased ? s->s1b : s->s1a;
    for (int w = 0; w < C; ++w) MCPM_TRY(xfer_wait(p, h1[w]));
    MCPM_TRY(mcpm_slab_xfused(p, x_in, s3a, 0));                      // -> A, G
    const std::vector<int> order = edge_first(C);
    std::vector<int> hA(C, -1), hG(C, -1);
    for (int w : order) {
        bool al2;
        MCPM_TRY(a2a_chunk(c, s3b, s3a, w, &hA[w], &aliased));
        MCPM_TRY(a2a_chunk(c, c.spec(s3b, 1), c.spec(s3a, 1), w, &hG[w], &al2));
    }
    float *src = aliased ? s3a : s3b, *dst = aliased ? s3b : s3a;
    parts = chunk_parts(p, c.d, true);
    int n_edge_chunks = 0;
    for (int w : order) n_edge_chunks += parts[w].edges.empty() ? 0 : 1;
    int fill = -1;
    bool filling = false;
    const int cw = c.nxl / C;
    for (int w : order) {
        MCPM_TRY(xfer_wait(p, hA[w]));
        MCPM_TRY(c.win(Win{w * cw, cw}));
        MCPM_TRY(mcpm_slab_ycol2(p, src, dst, 1, 1, 0, 1));          // A -> force spectrum 0
        MCPM_TRY(xfer_wait(p, hG[w]));
        MCPM_TRY(mcpm_slab_ycol2(p, src, dst, 1, 1, 0, 2));          // G -> force spectra 1, 2
        for (const Win &e : parts[w].edges) {                        // the edge planes first ...
            MCPM_TRY(c.win(e));
            MCPM_TRY(mcpm_slab_zinv3_il(p, dst, c.interior_il(f3_il)));
        }
        if (!parts[w].edges.empty() && --n_edge_chunks == 0) {       // ... so that the ghost fill runs under the rest
            MCPM_TRY(halo_fill_begin(c, f3_il, 1, 0, 3 * c.plane, &fill));
            filling = true;
        }
        if (parts[w].has_inner) {
            MCPM_TRY(c.win(parts[w].inner));
            MCPM_TRY(mcpm_slab_zinv3_il(p, dst, c.interior_il(f3_il)));
        }
    }
    MCPM_TRY(c.win_all());
    if (filling) MCPM_TRY(xfer_wait(p, fill));
    return MCPM_OK;
}

// three cotangent meshes (ghost add possibly in flight) -> rho_bar with its ghost planes filled (SlabPM.force_meshes_vjp_gen)
int force_meshes_vjp(const Ctx &c, float *fbar3_ext, float *rhobar_ext, bool add_pending, int add_ticket) {
    mcpm_plan *p = c.p;
    SlabState *s = c.s;
    const int C = c.C;
    float *s3a = s->s6a, *s3b = s->s6b;
    std::vector<Part> parts = chunk_parts(p, c.d, add_pending);
    std::vector<int> ha(C, -1), hb(C, -1);
    bool aliased = false;
    auto zy = [&](Win w) -> int {   // three z transforms, then a = FFTy(f_bar_x) and b = ky FFTy(f_bar_y) + kz FFTy(f_bar_z)
        MCPM_TRY(c.win(w));
        for (int k = 0; k < 3; ++k) MCPM_TRY(mcpm_slab_zfwd(p, c.interior(fbar3_ext + k * c.Me), c.Me, c.spec(s3a, k), 1));
        return mcpm_slab_ycol2(p, s3a, s3b, 0, 0, 1, 3);
    };
    auto send = [&](int w) -> int {
        bool al2;
        MCPM_TRY(a2a_chunk(c, c.spec(s->s6a, 3), s3b, w, &ha[w], &aliased));
        return a2a_chunk(c, c.spec(s->s6a, 4), c.spec(s3b, 1), w, &hb[w], &al2);
    };
    for (int w = 0; w < C; ++w) {
        if (parts[w].has_inner) MCPM_TRY(zy(parts[w].inner));
        if (parts[w].edges.empty()) MCPM_TRY(send(w));
    }
    if (add_pending) MCPM_TRY(halo_add_finish(c, fbar3_ext, 3, c.Me, add_ticket));
    for (int w = 0; w < C; ++w) {
        for (const Win &e : parts[w].edges) MCPM_TRY(zy(e));
        if (!parts[w].edges.empty()) MCPM_TRY(send(w));
    }
    MCPM_TRY(c.win_all());
    for (int w = 0; w < C; ++w) MCPM_TRY(xfer_wait(p, ha[w]));
    for (int w = 0; w < C; ++w) MCPM_TRY(xfer_wait(p, hb[w]));
    // (a, b) received side by side in s6a[3 ss : 5 ss], or still in s3b[0 : 2 ss] when the transpose was aliased
    float *x_in = aliased ? s3b : c.spec(s->s6a, 3);
    MCPM_TRY(mcpm_slab_xfused(p, x_in, s->s1a, 1));
    const std::vector<int> order = edge_first(C);
    std::vector<int> h1(C, -1);
    for (int w : order) MCPM_TRY(a2a_chunk(c, s->s1b, s->s1a, w, &h1[w], &aliased));
    float *y_in = aliased ? s->s1a : s->s1b, *y_out = aliased ? s->s1b : s->s1a;
    parts = chunk_parts(p, c.d, true);
    int n_edge_chunks = 0;
    for (int w : order) n_edge_chunks += parts[w].edges.empty() ? 0 : 1;
    int fill = -1;
    bool filling = false;
    const int cw = c.nxl / C;
    for (int w : order) {
        MCPM_TRY(xfer_wait(p, h1[w]));
        MCPM_TRY(c.win(Win{w * cw, cw}));
        MCPM_TRY(mcpm_slab_ycol(p, y_in, y_out, 1, +1, 1, 0));
        for (const Win &e : parts[w].edges) {
            MCPM_TRY(c.win(e));
            MCPM_TRY(mcpm_slab_zinv(p, y_out, c.interior(rhobar_ext), c.Me, 1));
        }
        if (!parts[w].edges.empty() && --n_edge_chunks == 0) {
            MCPM_TRY(halo_fill_begin(c, rhobar_ext, 1, 0, c.plane, &fill));
            filling = true;
        }
        if (parts[w].has_inner) {
            MCPM_TRY(c.win(parts[w].inner));
            MCPM_TRY(mcpm_slab_zinv(p, y_out, c.interior(rhobar_ext), c.Me, 1));
        }
    }
    MCPM_TRY(c.win_all());
    if (filling) MCPM_TRY(xfer_wait(p, fill));
    return MCPM_OK;
}

}  // namespace

void mcpm_slab_state_free(mcpm_plan *p) {
    SlabState *s = state(p);
    if (!s) return;
    if (s->comm) (void)rccl_api()->CommDestroy(s->comm);
    if (s->cs) (void)hipStreamDestroy(s->cs);
    if (s->events)
        for (int i = 0; i < NTICKETS; ++i) {
            (void)hipEventDestroy(s->ready[i]);
            (void)hipEventDestroy(s->done[i]);
        }
    if (s->dmax_dev) (void)hipFree(s->dmax_dev);
    if (s->dmax_host) (void)hipHostFree(s->dmax_host);
    for (int i = 0; i < DMAX_RING; ++i)
        if (s->dmax_ev[i]) (void)hipEventDestroy(s->dmax_ev[i]);
    delete s;
    p->slab_state = nullptr;
}

extern "C" {

int mcpm_slab_rccl_unique_id(void *id128) {
    if (!id128) return MCPM_E_ARG;
    RcclApi *a = rccl_api();
    if (!a->lib) return mcpm_fail(nullptr, MCPM_E_RCCL, a->why);
    ncclUniqueId id;
    const ncclResult_t r = a->GetUniqueId(&id);
    if (r != ncclSuccess) return mcpm_fail(nullptr, MCPM_E_RCCL, std::string("ncclGetUniqueId: ") + a->GetErrorString(r));
    std::memcpy(id128, &id, sizeof(id));
    return MCPM_OK;
}

int mcpm_slab_comm_shutdown(mcpm_plan *p) {
    if (!p) return MCPM_E_ARG;
    MCPM_HIP(p, hipStreamSynchronize(p->stream));
    mcpm_slab_state_free(p);
    return MCPM_OK;
}

static SlabState *fresh_state(mcpm_plan *p) {
    if (p->slab_state) mcpm_slab_state_free(p);
    SlabState *s = new (std::nothrow) SlabState();
    p->slab_state = s;
    return s;
}

int mcpm_slab_comm_init_local(mcpm_plan *p) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, p->nranks == 1, MCPM_E_ARG, "the local transport serves one rank");
    SlabState *s = fresh_state(p);
    if (!s) return mcpm_fail(p, MCPM_E_NOMEM, "host allocation");
    s->mode = MODE_LOCAL;
    return MCPM_OK;
}

int mcpm_slab_comm_init_rccl(mcpm_plan *p, const void *id128) {
    if (!p || !id128) return MCPM_E_ARG;
    RcclApi *a = rccl_api();
    if (!a->lib) return mcpm_fail(p, MCPM_E_RCCL, a->why);
    SlabState *s = fresh_state(p);
    if (!s) return mcpm_fail(p, MCPM_E_NOMEM, "host allocation");
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    RCCL_TRY(p, a->CommInitRank(&s->comm, p->nranks, id, p->rank));
    // A high-priority stream: HIP multiplexes streams onto a few hardware queues, and streams that share a queue run in
    // submission order -- on the compute stream's queue the transfers would never overlap a kernel.
    // Priority streams get queues of their own, and the copy kernels of an exchange are short.  MCPM_SLAB_COMM_PRIORITY=0: default.
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    const char *pe = getenv("MCPM_SLAB_COMM_PRIORITY");
    if (pe && atoi(pe) == 0) MCPM_HIP(p, hipStreamCreateWithFlags(&s->cs, hipStreamNonBlocking));
    else MCPM_HIP(p, hipStreamCreateWithPriority(&s->cs, hipStreamNonBlocking, hi));
    MCPM_TRY(ensure_events(p, s));
    s->mode = MODE_RCCL;
    return MCPM_OK;
}

int mcpm_slab_comm_init_ops(mcpm_plan *p, const mcpm_comm_ops *ops) {
    if (!p || !ops) return MCPM_E_ARG;
    MCPM_REQUIRE(p, ops->p2p_begin && ops->wait && ops->allreduce_max_f32, MCPM_E_ARG, "mcpm_comm_ops: null callback");
    SlabState *s = fresh_state(p);
    if (!s) return mcpm_fail(p, MCPM_E_NOMEM, "host allocation");
    s->ops = *ops;
    s->mode = MODE_OPS;
    return MCPM_OK;
}

// One small round of every exchange pattern a slab step uses, checked on the host: an equal-split all-to-all (the FFT transposes),
// a send / receive with both x neighbours (the ghost planes) and the max all-reduce (the ghost depth).  A communicator that came up
// but cannot move data between its peers -- the first contact of this code with a multi-GPU node is the driver's scaling run --
// fails HERE, with a message, and the host can still choose the torch.distributed transport (dist.SlabPM); a failure inside the
// first step could only abort the run.  Collective: every rank of the plan's communicator calls it.  Synchronises the stream.
int mcpm_slab_comm_selftest(mcpm_plan *p) {
    if (!p) return MCPM_E_ARG;
    SlabState *s = state(p);
    MCPM_REQUIRE(p, s && s->mode != MODE_NONE, MCPM_E_ARG, "mcpm_slab_comm_selftest: no transport (mcpm_slab_comm_init_*)");
    const int P = p->nranks, r = p->rank, left = (r - 1 + P) % P, right = (r + 1) % P;
    constexpr int W = 16;                                  // floats per block
    const size_t nblk = (size_t)P + 2, bytes = nblk * W * sizeof(float);
    std::vector<float> h(nblk * W), back(nblk * W, -1.f);
    for (int q = 0; q < P; ++q)
        for (int i = 0; i < W; ++i) h[(size_t)q * W + i] = (float)(1000 * r + 10 * q) + 0.25f * i;      // block q: from r to q
    for (int i = 0; i < W; ++i) h[(size_t)P * W + i] = (float)(-1000 * r - 1) - 0.25f * i;              // to the left neighbour
    for (int i = 0; i < W; ++i) h[(size_t)(P + 1) * W + i] = (float)(-1000 * r - 2) - 0.25f * i;        // to the right neighbour
    float *src = nullptr, *dst = nullptr, *mx = nullptr;
    MCPM_HIP(p, hipMalloc((void **)&src, bytes));
    MCPM_HIP(p, hipMalloc((void **)&dst, bytes));
    MCPM_HIP(p, hipMalloc((void **)&mx, sizeof(float)));
    int rc = MCPM_OK;
    auto run = [&]() -> int {
        MCPM_HIP(p, hipMemcpyAsync(src, h.data(), bytes, hipMemcpyHostToDevice, p->stream));
        MCPM_HIP(p, hipMemsetAsync(dst, 0xff, bytes, p->stream));
        const float mine = (float)(r + 1);
        MCPM_HIP(p, hipMemcpyAsync(mx, &mine, sizeof(float), hipMemcpyHostToDevice, p->stream));
        int t0 = -1, t1 = -1;
        {
            Batch b;
            for (int q = 0; q < P; ++q) b.send(src + (size_t)q * W, W * sizeof(float), q);
            for (int q = 0; q < P; ++q) b.recv(dst + (size_t)q * W, W * sizeof(float), q);
            b.a2a_in = src, b.a2a_out = dst, b.a2a_bytes = W * sizeof(float);
            MCPM_TRY(xfer_begin(p, b, &t0));
        }
        {   // the order of halo_fill_begin: to left, to right; from right, from left
            Batch b;
            b.send(src + (size_t)P * W, W * sizeof(float), left);
            b.send(src + (size_t)(P + 1) * W, W * sizeof(float), right);
            b.recv(dst + (size_t)(P + 1) * W, W * sizeof(float), right);
            b.recv(dst + (size_t)P * W, W * sizeof(float), left);
            MCPM_TRY(xfer_begin(p, b, &t1));
        }
        MCPM_TRY(xfer_wait(p, t0));
        MCPM_TRY(xfer_wait(p, t1));
        if (s->mode == MODE_RCCL && P > 1) {
            RcclApi *a = rccl_api();
            const int t = s->next_ticket;
            s->next_ticket = (t + 1) % NTICKETS;
            MCPM_HIP(p, hipEventRecord(s->ready[t], p->stream));
            MCPM_HIP(p, hipStreamWaitEvent(s->cs, s->ready[t], 0));
            RCCL_TRY(p, a->AllReduce(mx, mx, 1, ncclFloat32, ncclMax, s->comm, s->cs));
            MCPM_HIP(p, hipEventRecord(s->done[t], s->cs));
            MCPM_HIP(p, hipStreamWaitEvent(p->stream, s->done[t], 0));
        } else if (s->mode == MODE_OPS) {
            if (s->ops.allreduce_max_f32(s->ops.ctx, mx, (void *)p->stream) != 0)
                return mcpm_fail(p, MCPM_E_RCCL, "transport callback allreduce_max_f32 failed");
        }
        float got = 0.f;
        MCPM_HIP(p, hipMemcpyAsync(back.data(), dst, bytes, hipMemcpyDeviceToHost, p->stream));
        MCPM_HIP(p, hipMemcpyAsync(&got, mx, sizeof(float), hipMemcpyDeviceToHost, p->stream));
        MCPM_HIP(p, hipStreamSynchronize(p->stream));
        for (int q = 0; q < P; ++q)
            for (int i = 0; i < W; ++i)
                if (back[(size_t)q * W + i] != (float)(1000 * q + 10 * r) + 0.25f * i)
                    return mcpm_fail(p, MCPM_E_RCCL, "transport self-test: all-to-all block from rank " + std::to_string(q) + " is wrong");
        // what the left neighbour sent to ITS right (block P+1 of rank `left`) lands in my from-left block P, and vice versa
        for (int i = 0; i < W; ++i) {
            if (back[(size_t)P * W + i] != (float)(-1000 * left - 2) - 0.25f * i)
                return mcpm_fail(p, MCPM_E_RCCL, "transport self-test: the block from the left neighbour is wrong");
            if (back[(size_t)(P + 1) * W + i] != (float)(-1000 * right - 1) - 0.25f * i)
                return mcpm_fail(p, MCPM_E_RCCL, "transport self-test: the block from the right neighbour is wrong");
        }
        if (got != (float)P) return mcpm_fail(p, MCPM_E_RCCL, "transport self-test: max all-reduce gave " + std::to_string(got));
        return MCPM_OK;
    };
    rc = run();
    (void)hipStreamSynchronize(p->stream);
    (void)hipFree(src);
    (void)hipFree(dst);
    (void)hipFree(mx);
    return rc;
}

int mcpm_slab_bind_workspace(mcpm_plan *p, float *rho, float *f3, float *s1a, float *s1b, float *s6a, float *s6b, float *Fb,
                             float *halo) {
    if (!p) return MCPM_E_ARG;
    SlabState *s = state(p);
    MCPM_REQUIRE(p, s != nullptr, MCPM_E_ARG, "mcpm_slab_bind_workspace: call mcpm_slab_comm_init_* first");
    MCPM_REQUIRE(p, rho && f3 && s1a && s1b && s6a && s6b && Fb && halo, MCPM_E_ARG, "mcpm_slab_bind_workspace: null buffer");
    s->rho = rho; s->f3 = f3; s->s1a = s1a; s->s1b = s1b; s->s6a = s6a; s->s6b = s6b; s->Fb = Fb; s->halo = halo;
    return MCPM_OK;
}

int mcpm_slab_step_f32(mcpm_plan *p, const float *x, const float *v, double alpha, double beta, double tau, int paint_order,
                       int depth, float *f3_out, float *x_out, float *v_out) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, x && v && f3_out && x_out && v_out, MCPM_E_ARG, "mcpm_slab_step_f32: null buffer");
    Ctx c;
    MCPM_TRY(make_ctx(p, depth, &c));
    MCPM_TRY(mcpm_paint_f32(p, x, p->Np, MCPM_POS_LATTICE, nullptr, 1, 1.f, paint_order, c.s->rho, 0));
    int add = -1;
    MCPM_TRY(halo_add_begin(c, c.s->rho, 1, 0, &add));
    MCPM_TRY(force_meshes(c, c.s->rho, f3_out, true, add));
    MCPM_TRY(mcpm_kick_drift_il_f32(p, x, v, p->Np, MCPM_POS_LATTICE, f3_out, paint_order, alpha, beta, tau, x_out, v_out));
    return dmax_enqueue(c);
}

int mcpm_slab_step_vjp_f32(mcpm_plan *p, const float *x, const float *v, const float *f3, double alpha, double beta, double tau,
                           int paint_order, int depth, float *xb, float *vb, double *alpha_bar, double *beta_bar,
                           double dtau_ddg, double *dg_bar, int has_next, double next_beta, double next_tau) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, x && v && f3 && xb && vb, MCPM_E_ARG, "mcpm_slab_step_vjp_f32: null buffer");
    Ctx c;
    MCPM_TRY(make_ctx(p, depth, &c));
    float *fb = nullptr;
    MCPM_TRY(mcpm_plan_chained_fb(p, beta, tau, xb, vb, &fb));
    if (!fb) {
        MCPM_TRY(mcpm_kick_f32(p, vb, xb, p->Np, (float)beta, (float)(beta * tau), c.s->Fb));
        fb = c.s->Fb;
    }
    MCPM_TRY(mcpm_paint3_f32(p, x, p->Np, MCPM_POS_LATTICE, fb, paint_order, c.s->f3, 0));
    int add = -1;
    MCPM_TRY(halo_add_begin(c, c.s->f3, 3, c.Me, &add));      // ONE exchange for the three components
    MCPM_TRY(force_meshes_vjp(c, c.s->f3, c.s->rho, true, add));
    if (has_next) MCPM_TRY(mcpm_plan_hint_next_adjoint(p, next_beta, next_tau));
    return mcpm_step_adjoint_particles_il_f32(p, x, v, f3, c.s->rho, alpha, beta, tau, paint_order, xb, vb, alpha_bar, beta_bar,
                                              dtau_ddg, dg_bar);
}

int64_t mcpm_slab_dmax_seq(const mcpm_plan *p) {
    const SlabState *s = p ? static_cast<const SlabState *>(p->slab_state) : nullptr;
    return s ? s->dmax_seq : 0;
}

int mcpm_slab_dmax_read(mcpm_plan *p, int64_t seq, float *value, int *valid) {
    if (!p || !value || !valid) return MCPM_E_ARG;
    SlabState *s = state(p);
    *valid = 0;
    if (!s || seq < 0 || seq >= s->dmax_seq || seq < s->dmax_seq - DMAX_RING) return MCPM_OK;   // never taken, or overwritten
    const int slot = (int)(seq % DMAX_RING);
    MCPM_HIP(p, hipEventSynchronize(s->dmax_ev[slot]));
    *value = s->dmax_host[slot];
    *valid = 1;
    return MCPM_OK;
}

}  // extern "C"
