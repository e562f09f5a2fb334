"""
x-slab decomposition of the PM step over the GPUs of one node (no counterpart in the reference,
whose only multi-device mode is independent chains, script.py:13-20).

Decomposition (one process per GPU, `torch.distributed`, backend "nccl" = RCCL over xGMI):
  * rank r owns mesh planes [r nx/P, (r+1) nx/P) and -- Lagrangian ownership -- the lattice particles of those
    planes.  Particles are stored as displacements from their lattice point, so they NEVER migrate; what
    crosses slab boundaries is mass, through `ghost` extra planes on each side of the local mesh.
  * paint -> ghost planes added into the neighbours' interiors (point-to-point isend / irecv of contiguous plane runs);
    FFT: local z and y passes -> ONE all-to-all (x <-> y transpose, written directly in transposed order by the
    y pass) -> fused x pass (x FFT . k-space . inverse x FFT) -> all-to-all back (two spectra) -> local y pass, ONE z pass
    that writes the three force components interleaved [x][y][z][3];
    read: interior planes sent straight into the neighbours' ghost planes (point-to-point, no staging copy: the three
    components of a plane run are one contiguous block), then the fused read+kick+drift.
  The all-to-all is used for the FFT transpose only; ghost planes are point-to-point.
The adjoint mirrors it (3 weighted paints + ghost add, 3 forward / 1 inverse transform, ghost fill, fused
adjoint particle kernel); scalar cotangents are all-reduced once at the end.

The LPT start and its adjoint are slab-decomposed as well (`SlabPM.lpt`, `SlabPM.lpt_vjp`): the replicated initial
half-spectrum is already in k-space, so each rank runs the spectrum-side x pass on its y rows, and the same
all-to-all / y / z passes follow; the init_mesh cotangent is assembled with one all-reduce.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import nbody
from ._lib import lib, check, POS_LATTICE


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _exhaust(gen):
    for _ in gen:
        pass


class _Done:
    """Handle of an exchange that has already completed."""
    done = True

    def wait(self):
        pass


class _HandleOf:
    """Adapts anything with .wait() (a _Works, a _Done) to the torch Work interface _Works expects."""

    def __init__(self, h):
        self.h = h

    def wait(self):
        self.h.wait()


class _Works:
    def __init__(self, works, after=None):
        self.works, self.after, self.done = works, after, False

    def wait(self):
        if self.done:      # several consumers may wait on one exchange (the three components of a batched ghost add)
            return
        self.done = True
        for w in self.works:
            w.wait()
        if self.after:
            self.after()


def _pinned_slot(owner):
    """A (pinned one-float tensor, event) pair from a small ring owned by the communicator: a pinned allocation per step
    costs more host time than the step's launches."""
    ring = getattr(owner, "_pin_ring", None)
    if ring is None:
        ring = owner._pin_ring = [[torch.empty(1, dtype=torch.float32, pin_memory=True), torch.cuda.Event()] for _ in range(4)]
        owner._pin_next = 0
    slot = ring[owner._pin_next % len(ring)]
    owner._pin_next += 1
    return slot[0], slot[1]


class LocalComm:
    """Single-rank communicator: ghost exchanges are local copies (the periodic wrap of the own planes); a one-rank
    all-to-all is the identity, so `SlabPM._a2a` hands the input buffer on instead of copying it (`alias_a2a`)."""
    world, rank = 1, 0
    alias_a2a = True

    def all_to_all(self, out, inp, async_op=False):
        out.copy_(inp)
        return _Done()

    def neighbour_exchange(self, to_left, to_right, from_left, from_right, async_op=False):
        from_right.copy_(to_left)   # my left neighbour is me: what I send left arrives "from the right"
        from_left.copy_(to_right)
        return _Done()

    def all_reduce_sum(self, t):
        return t

    def all_reduce_max_float(self, v):
        return float(v)

    def all_reduce_max_async(self, v):
        """Maximum over ranks of one device scalar without stopping the host: returns (pinned host tensor, event); the
        value is in the tensor once the event has completed."""
        host, ev = _pinned_slot(self)
        host.copy_(v.detach().reshape(1).float(), non_blocking=True)
        ev.record()
        return host, ev

    def all_gather_cat(self, t):
        return t


class TorchComm:
    """torch.distributed communicator.  With the "gloo" backend (CPU tests, or several ranks sharing one GPU in
    the single-GPU test box) device tensors are staged through host memory; with "nccl" (RCCL) they are used
    in place."""

    def __init__(self, group=None, stage=None):
        import torch.distributed as td
        self.td, self.group = td, group
        self.world, self.rank = td.get_world_size(group), td.get_rank(group)
        # stage=False with gloo and CPU tensors runs the un-staged (RCCL) code path of the exchanges in the CPU tests
        self.stage = (td.get_backend(group) == "gloo") if stage is None else bool(stage)

    def _h(self, t):
        return t.cpu() if (self.stage and t.is_cuda) else t

    def all_to_all(self, out, inp, async_op=False):
        """Equal-split all-to-all along dim 0.  With async_op the RCCL kernel runs on its own stream, ordered after
        the work already enqueued on the current stream; `.wait()` orders the current stream after it."""
        if self.stage and inp.is_cuda:
            o = torch.empty(out.shape, dtype=out.dtype)
            self.td.all_to_all_single(o.view(torch.float32) if o.is_complex() else o,
                                      (inp.cpu().view(torch.float32) if inp.is_complex() else inp.cpu()), group=self.group)
            out.copy_(o)
            return _Done()
        vo = torch.view_as_real(out) if out.is_complex() else out
        vi = torch.view_as_real(inp) if inp.is_complex() else inp
        w = self.td.all_to_all_single(vo, vi, group=self.group, async_op=async_op)
        return _Works([w]) if async_op else _Done()

    def neighbour_exchange(self, to_left, to_right, from_left, from_right, async_op=False):
        """Point-to-point exchange with the two x neighbours (one batched isend / irecv group).  With RCCL the tensors are
        used in place: callers pass contiguous plane runs of the meshes themselves, so nothing is staged or copied.  (An
        earlier version sent both messages through one all_to_all_single with split sizes to save ~45 us of host time
        per exchange; the all-to-all is reserved for the FFT transpose, and the host is not the bottleneck.)"""
        td, P, r = self.td, self.world, self.rank
        left, right = (r - 1) % P, (r + 1) % P
        sl, sr = self._h(to_left), self._h(to_right)
        rl = torch.empty(from_left.shape, dtype=from_left.dtype) if (self.stage and from_left.is_cuda) else from_left
        rr = torch.empty(from_right.shape, dtype=from_right.dtype) if (self.stage and from_right.is_cuda) else from_right
        # with two ranks both messages go to the same peer: my first send (to the left) must meet the peer's first
        # receive, which is therefore its "from the right"
        ops = [td.P2POp(td.isend, sl, left, self.group), td.P2POp(td.isend, sr, right, self.group),
               td.P2POp(td.irecv, rr, right, self.group), td.P2POp(td.irecv, rl, left, self.group)]
        works = td.batch_isend_irecv(ops)

        def finish():
            if rl is not from_left:
                from_left.copy_(rl)
            if rr is not from_right:
                from_right.copy_(rr)

        h = _Works(works, finish)
        if async_op and not self.stage:
            return h
        h.wait()
        return _Done()

    def all_reduce_sum(self, t):
        if t.is_complex():
            return torch.view_as_complex(self.all_reduce_sum(torch.view_as_real(t)))
        h = self._h(t).clone()
        self.td.all_reduce(h, group=self.group)
        return h.to(t.device)

    def all_reduce_max_float(self, v):
        """Maximum over ranks of one device (or host) scalar, returned as a python float (synchronises the host)."""
        t = v.detach().reshape(1).float() if isinstance(v, torch.Tensor) else torch.tensor([float(v)])
        t = t.cpu() if self.stage else t.to(torch.device("cuda", torch.cuda.current_device()))
        self.td.all_reduce(t, op=self.td.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def all_reduce_max_async(self, v):
        """See LocalComm.all_reduce_max_async.  RCCL: the all-reduce runs on the communicator's stream, the current stream
        (not the host) waits for it and copies the result to pinned memory.  Staged (gloo) communicators have to stop the
        host anyway: the value is returned with no event."""
        if self.stage:
            return torch.tensor([self.all_reduce_max_float(v)], dtype=torch.float32), None
        t = v.detach().reshape(1).float().clone()
        self.td.all_reduce(t, op=self.td.ReduceOp.MAX, group=self.group, async_op=True).wait()
        host, ev = _pinned_slot(self)
        host.copy_(t, non_blocking=True)
        ev.record()
        return host, ev

    def all_gather_cat(self, t):
        h = self._h(t).contiguous()
        outs = [torch.empty_like(h) for _ in range(self.world)]
        self.td.all_gather(outs, h, group=self.group)
        return torch.cat(outs, dim=0).to(t.device)


class _CommOps(C.Structure):
    """include/mcpm.h `mcpm_comm_ops`."""
    P2P = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int), C.c_int,
                      C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int))
    WAIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)
    AMAX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
    _fields_ = [("ctx", C.c_void_p), ("p2p_begin", P2P), ("wait", WAIT), ("allreduce_max_f32", AMAX)]


class HostStagedOps:
    """Transport callbacks for the native slab steps (`mcpm_slab_comm_init_ops`) over a STAGED torch.distributed communicator
    (gloo): every batch is completed synchronously through host memory.  Test infrastructure: it lets several ranks sharing one
    GPU (or CPU-only gloo runs of the algebra) drive the multi-rank code of csrc/slab.hip; production uses the plan-owned RCCL
    communicator."""

    def __init__(self, comm):
        import torch.distributed as td
        self.td, self.comm = td, comm
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipMemcpy.restype = C.c_int
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.error = None
        self.struct = _CommOps(None, _CommOps.P2P(self._p2p), _CommOps.WAIT(self._wait), _CommOps.AMAX(self._amax))

    def _guard(self, fn, *a):
        try:
            return fn(*a)
        except Exception as e:      # an exception must not cross the C frame: reported as a failed callback
            self.error = e
            return 1

    def _copy(self, dst, src, n, kind):
        if self.hip.hipMemcpy(dst, src, n, kind) != 0:
            raise RuntimeError("hipMemcpy failed in the staged transport")

    def _p2p(self, ctx, ns, sp, sb, speer, nr, rp, rb, rpeer, stream, ticket):
        return self._guard(self._p2p_impl, ns, sp, sb, speer, nr, rp, rb, rpeer, stream, ticket)

    def _p2p_impl(self, ns, sp, sb, speer, nr, rp, rb, rpeer, stream, ticket):
        td, me, grp = self.td, self.comm.rank, self.comm.group
        self.hip.hipStreamSynchronize(stream)
        ops, recvs, self_s, self_r = [], [], [], []
        for i in range(ns):
            if speer[i] == me:
                self_s.append((sp[i], sb[i]))
                continue
            t = torch.empty(sb[i], dtype=torch.uint8)
            self._copy(t.data_ptr(), sp[i], sb[i], 2)                 # device -> host
            ops.append(td.P2POp(td.isend, t, speer[i], grp))
        for i in range(nr):
            if rpeer[i] == me:
                self_r.append((rp[i], rb[i]))
                continue
            t = torch.empty(rb[i], dtype=torch.uint8)
            recvs.append((rp[i], t))
            ops.append(td.P2POp(td.irecv, t, rpeer[i], grp))
        assert len(self_s) == len(self_r)
        for (s_, n_), (r_, m_) in zip(self_s, self_r):                # own block of an all-to-all: k-th send meets k-th receive
            assert n_ == m_
            if s_ != r_:
                self._copy(r_, s_, n_, 3)                             # device -> device
        if ops:
            for w in td.batch_isend_irecv(ops):
                w.wait()
        for dst, t in recvs:
            self._copy(dst, t.data_ptr(), t.numel(), 1)               # host -> device
        ticket[0] = 0
        return 0

    def _wait(self, ctx, ticket, stream):
        return 0

    def _amax(self, ctx, dev, stream):
        return self._guard(self._amax_impl, dev, stream)

    def _amax_impl(self, dev, stream):
        self.hip.hipStreamSynchronize(stream)
        t = torch.empty(1, dtype=torch.float32)
        self._copy(t.data_ptr(), dev, 4, 2)
        self.td.all_reduce(t, op=self.td.ReduceOp.MAX, group=self.comm.group)
        self._copy(dev, t.data_ptr(), 4, 1)
        return 0


class HaloMixin:
    """Ghost-plane algebra of an x-slab (needs self.comm, self.G, self.nxl); meshes are (..., nxl + 2G, ny, nz)."""

    def depth(self):
        """Ghost planes the exchanges cover: all G, or the `self.ge` planes next to the interior that this step's
        displacements can reach (SlabPM.set_depth)."""
        return min(self.G, getattr(self, "ge", self.G))

    def halo_add(self, ext, async_op=False):
        """Adds my ghost planes into the neighbours' interiors (after a paint).  With async_op returns a handle
        whose wait() completes the exchange and does the additions."""
        G, nxl, d = self.G, self.nxl, self.depth()
        lo = ext[..., G - d:G, :, :].contiguous()
        hi = ext[..., G + nxl:G + nxl + d, :, :].contiguous()
        from_l, from_r = torch.empty_like(hi), torch.empty_like(lo)
        h = self.comm.neighbour_exchange(lo, hi, from_l, from_r, async_op=async_op)

        def finish():
            h.wait()
            ext[..., G:G + d, :, :] += from_l                  # the left neighbour's high ghost = my lowest interior planes
            ext[..., G + nxl - d:G + nxl, :, :] += from_r      # the right neighbour's low ghost = my highest interior planes

        if async_op:
            return _Works([], finish)
        finish()
        return _Done()

    def halo_fill(self, ext, async_op=False):
        """Fills my ghost planes from the neighbours' interiors (before a read).  With async_op returns a handle
        whose wait() completes the exchange and writes the ghosts (lets the next component's FFT passes overlap)."""
        G, nxl, d = self.G, self.nxl, self.depth()
        to_l = ext[..., G:G + d, :, :].contiguous()                  # becomes the left neighbour's high ghost
        to_r = ext[..., G + nxl - d:G + nxl, :, :].contiguous()      # becomes the right neighbour's low ghost
        from_l, from_r = torch.empty_like(to_r), torch.empty_like(to_l)
        h = self.comm.neighbour_exchange(to_l, to_r, from_l, from_r, async_op=async_op)

        def finish():
            h.wait()
            ext[..., G - d:G, :, :] = from_l
            ext[..., G + nxl:G + nxl + d, :, :] = from_r

        if async_op:
            return _Works([], finish)
        finish()
        return _Done()


class PlaneHalo:
    """Ghost-plane exchanges on meshes whose LEADING axis is x: (nxl + 2G, ...).  A run of planes is then one contiguous
    block whatever follows (ny, nz) or (ny, nz, 3): the fill receives straight into the ghost planes and sends straight from
    the interior (no staging); the add receives into a scratch run and adds it."""

    def halo_fill_x(self, ext, async_op=False):
        G, nxl, d = self.G, self.nxl, self.depth()
        h = self.comm.neighbour_exchange(ext[G:G + d], ext[G + nxl - d:G + nxl], ext[G - d:G], ext[G + nxl:G + nxl + d],
                                         async_op=async_op)
        return h if async_op else _Done()

    def halo_add_x(self, ext, async_op=False):
        G, nxl, d = self.G, self.nxl, self.depth()
        from_l, from_r = torch.empty_like(ext[:d]), torch.empty_like(ext[:d])
        h = self.comm.neighbour_exchange(ext[G - d:G], ext[G + nxl:G + nxl + d], from_l, from_r, async_op=async_op)

        def finish():
            h.wait()
            ext[G:G + d] += from_l
            ext[G + nxl - d:G + nxl] += from_r

        if async_op:
            return _Works([], finish)
        finish()
        return _Done()


class SlabPM(HaloMixin, PlaneHalo):
    """Slab-decomposed PM stepper for one rank."""

    def __init__(self, mesh_shape, comm=None, ghost=16, device=None, adaptive_ghost=True, chunks=None, native=None):
        """chunks: every FFT transpose is split into this many all-to-alls of 1 / chunks of a spectrum each (chunks of
        nx_local / chunks planes, `mcpm_slab_set_chunks`), issued as soon as the chunk's z / y passes are done and consumed
        chunk by chunk on the other side, so that the transfers run under the passes of the other chunks.  Default
        (None): MCPM_SLAB_CHUNKS, else 1 on one rank, 4 with at least 128 local planes and 2 below (each chunk costs a
        launch of every z / y pass: +0.12 / +0.24 / +0.77 ms per step at 512^3 for 2 / 4 / 8 chunks, tools/chunk_sweep.sh)."""
        self.adaptive_ghost = bool(adaptive_ghost)
        self.comm = comm if comm is not None else LocalComm()
        P, r = self.comm.world, self.comm.rank
        nx, ny, nz = (int(s) for s in mesh_shape)
        if nx % P or ny % P:
            raise ValueError("nx and ny must be divisible by the number of ranks")
        self.shape, self.P, self.rank = (nx, ny, nz), P, r
        self.nxl, self.nyl, self.G = nx // P, ny // P, int(ghost)
        self.nxe = self.nxl + 2 * self.G
        self.device = device if device is not None else nbody._device()
        self.stream = torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        check(lib.mcpm_plan_create_slab(nx, ny, nz, P, r, self.G, C.c_void_p(self.stream.cuda_stream), C.byref(h)), None,
              "mcpm_plan_create_slab")
        self.h = h
        if chunks is None:
            import os
            chunks = int(os.environ.get("MCPM_SLAB_CHUNKS", "1" if P == 1 else ("4" if self.nxl >= 128 else "2")))
        while chunks > 1 and (self.nxl % chunks or self.nxl // chunks < 2 * self.G):    # a chunk holds at most one edge
            chunks //= 2
        self.chunks = max(1, int(chunks))
        check(lib.mcpm_slab_set_chunks(h, self.chunks), h, "mcpm_slab_set_chunks")
        self.Nl = self.nxl * ny * nz                   # local particles
        self.Me = self.nxe * ny * nz                   # ghost-extended local mesh
        self.plane = ny * nz
        ss = self.ss = lib.mcpm_slab_spec_elems(h)
        f32 = dict(dtype=torch.float32, device=self.device)
        c64 = dict(dtype=torch.complex64, device=self.device)
        self.rho = torch.zeros((self.nxe, ny, nz), **f32)
        self.f3 = torch.zeros((3, self.nxe, ny, nz), **f32)           # three separate meshes: lpt, cotangent paints
        self.f3il = None                                              # (nxe, ny, nz, 3) interleaved force mesh, on first use
        # kick_drift leaves max |d_x| of its output here (float bits in 64 slots, stride 32): the next step's ghost depth
        self.dmax = torch.zeros(64 * 32, dtype=torch.int32, device=self.device)
        # token of the positions whose max |d_x| the slots hold: (tensor kept alive, so its address cannot be reused;
        # torch version counter; kick-drift generation of this plan).  Anything else falls back to a reduction pass.
        self._dmax_token = None
        self._kd_gen = 0
        check(lib.mcpm_plan_track_dmax(h, C.c_void_p(self.dmax.data_ptr())), h, "mcpm_plan_track_dmax")
        self.s1a, self.s1b = torch.empty(ss, **c64), torch.empty(ss, **c64)
        self.s6a, self.s6b = torch.empty(6 * ss, **c64), torch.empty(6 * ss, **c64)     # lpt: 6 Hessian spectra
        self.s3a, self.s3b = self.s6a[:3 * ss], self.s6b[:3 * ss]
        self.h6 = None                                                                # (6, nxe, ny, nz), allocated by lpt
        self.Fb = torch.empty((self.Nl, 3), **f32)
        self.sbar = None
        # `native` (default: MCPM_SLAB_NATIVE, else on): step / step_vjp are ONE library call each (csrc/slab.hip issues the
        # kernels AND the exchanges, on a plan-owned RCCL communicator when the torch communicator is RCCL); off: the
        # exchanges are issued from here through torch.distributed, kernel by kernel (the reference both are tested against).
        import os
        asked = native is not None or "MCPM_SLAB_NATIVE" in os.environ
        if native is None:
            # Default: on wherever the library's transport has been exercised (one rank; several ranks through host callbacks),
            # OPT-IN (native=True / MCPM_SLAB_NATIVE=1) for RCCL with more than one rank: that branch of csrc/slab.hip has
            # never run on hardware (no multi-GPU box has been available to the builder), torch.distributed's RCCL path has.
            rccl_multi = not isinstance(self.comm, LocalComm) and not getattr(self.comm, "stage", False) and P > 1
            native = os.environ.get("MCPM_SLAB_NATIVE", "0" if rccl_multi else "1") != "0"
        self.native = bool(native)
        self.native_fallback = None        # why the library's transport was given up for torch.distributed, if it was
        if self.native:
            self._bring_up_native(strict=asked)

    def _bring_up_native(self, strict):
        """Communicator + one verified round of every exchange pattern (mcpm_slab_comm_selftest).  The ranks AGREE on the outcome:
        if any of them could not bring the library's transport up, all keep the torch.distributed path (native=False) -- unless
        the caller asked for the native path explicitly, in which case the error is raised."""
        err = None
        try:
            self._init_native_transport()
            check(lib.mcpm_slab_comm_selftest(self.h), self.h, "mcpm_slab_comm_selftest")
        except Exception as e:              # MCPM_E_RCCL, a missing librccl, a failing callback
            if self._host_ops is not None and self._host_ops.error is not None:
                e, self._host_ops.error = self._host_ops.error, None
            err = e
        bad = 0 if err is None else 1
        if self.comm.world > 1:
            bad = int(self.comm.all_reduce_max_float(float(bad)))
        if not bad:
            return
        if strict:
            raise err if err is not None else RuntimeError("another rank could not bring up the library's slab transport")
        lib.mcpm_slab_comm_shutdown(self.h)
        self.native = False
        self.native_fallback = repr(err) if err is not None else "another rank's transport failed"

    def _init_native_transport(self):
        h, comm = self.h, self.comm
        self._halo = torch.empty(6 * self.G * self.plane, dtype=torch.float32, device=self.device)
        self._host_ops = None
        if isinstance(comm, LocalComm):
            check(lib.mcpm_slab_comm_init_local(h), h, "mcpm_slab_comm_init_local")
        elif getattr(comm, "stage", False):                    # gloo: the host carries the bytes (tests)
            self._host_ops = HostStagedOps(comm)
            check(lib.mcpm_slab_comm_init_ops(h, C.byref(self._host_ops.struct)), h, "mcpm_slab_comm_init_ops")
        else:                                                   # RCCL: the plan gets its own communicator
            # Ordering against torch's communicator: every torch collective this class issues is waited for by the CURRENT
            # stream before anything else is enqueued on it (all_reduce_sum is stream-synchronous; all_reduce_max_async
            # calls .wait() on its work), and every library exchange starts behind an event recorded on that stream
            # (slab.hip xfer_begin) -- so a library exchange never runs beside an outstanding torch collective of this rank.
            td = comm.td
            # The ranks agree that EVERY one of them can open librccl before any of them enters a library collective
            # (ncclCommInitRank, the self-test): a rank that failed alone would leave the others waiting inside it.
            probe = torch.zeros(128, dtype=torch.uint8)
            mine = 0.0 if lib.mcpm_slab_rccl_unique_id(C.c_void_p(probe.data_ptr())) == 0 else 1.0
            if comm.all_reduce_max_float(mine) != 0.0:
                raise RuntimeError("librccl is missing or unusable on " + ("this rank" if mine else "another rank") + " (MCPM_E_RCCL)")
            idt = torch.zeros(129, dtype=torch.uint8)           # ncclUniqueId + "rank 0 has one" (a rank that raised before the
            if comm.rank == 0:                                  # broadcast would leave the others waiting in it)
                idt[128] = 1 if lib.mcpm_slab_rccl_unique_id(C.c_void_p(idt.data_ptr())) == 0 else 0
            dev_id = idt.to(self.device)                        # the group's backend moves device tensors
            src = td.get_global_rank(comm.group, 0) if comm.group is not None else 0
            td.broadcast(dev_id, src=src, group=comm.group)
            idt = dev_id.cpu()
            if int(idt[128]) != 1:
                raise RuntimeError("mcpm_slab_rccl_unique_id failed on rank 0 (librccl missing or unusable: MCPM_E_RCCL)")
            check(lib.mcpm_slab_comm_init_rccl(h, C.c_void_p(idt.data_ptr())), h, "mcpm_slab_comm_init_rccl")
        check(lib.mcpm_slab_bind_workspace(h, _p(self.rho), _p(self.f3), _p(self.s1a), _p(self.s1b), _p(self.s6a), _p(self.s6b),
                                           _p(self.Fb), _p(self._halo)), h, "mcpm_slab_bind_workspace")

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        try:
            if h:
                lib.mcpm_plan_destroy(h)
        except Exception:  # interpreter shutdown
            pass

    def call(self, name, *args):
        if name.startswith("mcpm_kick_drift"):     # every such call re-zeroes and refills the plan's dmax slots
            self._kd_gen += 1
            self._dmax_token = None
        check(getattr(lib, name)(self.h, *args), self.h, name)

    def _native_call(self, name, *args):
        rc = getattr(lib, name)(self.h, *args)
        if rc != 0 and self._host_ops is not None and self._host_ops.error is not None:
            err, self._host_ops.error = self._host_ops.error, None
            raise err
        check(rc, self.h, name)

    def out_of_ghost(self):
        """Cumulative count of deposits that fell beyond this rank's ghost planes (must stay 0)."""
        n = C.c_int64()
        self.call("mcpm_plan_slab_oob", C.byref(n))
        return n.value

    def _interior(self, ext, c=None):
        base = ext if c is None else ext[c]
        return C.c_void_p(base.data_ptr() + 4 * self.G * self.plane)

    def _interior_il(self, ext_il):
        return C.c_void_p(ext_il.data_ptr() + 4 * 3 * self.G * self.plane)

    def force_mesh_il(self):
        """Scratch interleaved force mesh (nxe, ny, nz, 3)."""
        if self.f3il is None:
            self.f3il = torch.zeros((self.nxe,) + self.shape[1:] + (3,), dtype=torch.float32, device=self.device)
        return self.f3il

    # ---- Poisson solve on slabs ----------------------------------------------------------------------------
    # Spectra buffers are component-major ([c][rank block][x_l][y_l][nzp]), so each force component is its own
    # all-to-all; they are issued asynchronously and the per-component y / z passes (and ghost fills) of one
    # component overlap the transfers of the next.
    def _spec(self, buf, c):
        return C.c_void_p(buf.data_ptr() + 8 * c * self.ss)

    def _win(self, w):
        self.call("mcpm_slab_set_window", int(w[0]), int(w[1]))

    # ---- chunked transposes ---------------------------------------------------------------------------------
    def _region(self, buf, w):
        """Chunk w of one spectrum buffer (ss complex): the contiguous [rank][x in chunk][y_local][nzp] region."""
        n = self.ss // self.chunks
        return buf[w * n:(w + 1) * n]

    def _a2a_chunk(self, out, inp, w, async_op=True):
        """All-to-all of chunk w of one spectrum; returns (handle, True if the result stayed in `inp`)."""
        if getattr(self.comm, "alias_a2a", False):
            return _Done(), True
        return self.comm.all_to_all(self._region(out, w), self._region(inp, w), async_op=async_op), False

    def _a2a_all(self, out, inp):
        """All chunks of one spectrum, asynchronously (the LPT transposes: not on the per-step path, no overlap scheduling).
        Unlike _a2a_chunk the result is always in `out` (a local communicator copies)."""
        n = self.ss // self.chunks
        return _Works([_HandleOf(self.comm.all_to_all(out[w * n:(w + 1) * n], inp[w * n:(w + 1) * n], async_op=True))
                       for w in range(self.chunks)])

    def _chunk_parts(self, pending):
        """Per chunk w: (interior planes (x0, n) or None, [edge plane runs]).  The first / last `depth` local planes are
        what a ghost exchange touches; with one pending they are transformed after it, the interior before."""
        d, nxl, cw = self.depth(), self.nxl, self.nxl // self.chunks
        out = []
        for w in range(self.chunks):
            lo, hi = w * cw, (w + 1) * cw
            if pending and nxl <= 2 * d:                 # every plane is within reach of the exchange
                out.append((None, [(lo, hi - lo)]))
                continue
            a = max(lo, d) if pending else lo
            b = min(hi, nxl - d) if pending else hi
            edges = ([(lo, a - lo)] if a > lo else []) + ([(b, hi - b)] if hi > b else [])
            out.append(((a, b - a) if b > a else None, edges))
        return out

    def _edge_first(self):
        """Chunk order of the return transposes: the chunks holding the edge planes first (their ghost fill then runs under
        the others)."""
        C = self.chunks
        return [0] + ([C - 1] if C > 1 else []) + list(range(1, C - 1))

    def force_meshes(self, *args, **kw):
        """See force_meshes_gen (this runs it to the end)."""
        _exhaust(self.force_meshes_gen(*args, **kw))

    def force_meshes_gen(self, rho_ext, f3_ext, fill_ghosts=True, rho_add=None, il=False):
        """GENERATOR: yields each time a collective has been launched and the next thing this trajectory would do is wait
        for it, so that a driver holding several independent trajectories can issue another one's kernels in between
        (bench.run_interleaved).  Interior of rho_ext -> the three force meshes (ghosts filled).  il=False: f3_ext is
        (3, nxe, ny, nz); il=True: ONE interleaved mesh (nxe, ny, nz, 3) (what the step kernels read: a CIC corner is one
        12-byte gather, and a run of ghost planes is one contiguous block for the three components).  `rho_add`: handle of
        the ghost add of rho_ext still in flight (None: ghosts already added).
        Transposes are chunked (`self.chunks`): a chunk's all-to-all leaves as soon as its z / y passes are done, and on the
        way back a chunk's inverse y / z passes run while the next chunks are still in flight.  Only two spectra (A, G)
        cross the second transpose: the y pass applies the y / z force factors."""
        ss, C = self.ss, self.chunks
        # ---- forward z / y passes chunk by chunk, each chunk's all-to-all right behind them
        parts = self._chunk_parts(rho_add is not None)
        h1, aliased = [None] * C, False

        def zy(win):
            self._win(win)
            self.call("mcpm_slab_zfwd", self._interior(rho_ext), self.Me, _p(self.s1a), 1)
            self.call("mcpm_slab_ycol", _p(self.s1a), _p(self.s1b), 1, -1, 0, 1)      # plain -> transposed order

        for w, (inner, edges) in enumerate(parts):
            if inner is not None:
                zy(inner)
            if not edges:
                h1[w], aliased = self._a2a_chunk(self.s1a, self.s1b, w)
        if rho_add is not None:
            yield
            rho_add.wait()
        for w, (inner, edges) in enumerate(parts):
            for e in edges:
                zy(e)
            if edges:
                h1[w], aliased = self._a2a_chunk(self.s1a, self.s1b, w)
        self._win((0, self.nxl))
        x_in = self.s1b if aliased else self.s1a
        yield
        for h in h1:
            h.wait()
        self.call("mcpm_slab_xfused", _p(x_in), _p(self.s3a), 0)                      # -> A, G
        # ---- back: A and G chunk by chunk, the edge chunks first
        order = self._edge_first()
        hA, hG = {}, {}
        for w in order:
            hA[w], aliased = self._a2a_chunk(self.s3b[:ss], self.s3a[:ss], w)
            hG[w], _ = self._a2a_chunk(self.s3b[ss:2 * ss], self.s3a[ss:2 * ss], w)
        src, dst = (self.s3a, self.s3b) if aliased else (self.s3b, self.s3a)
        parts = dict(enumerate(self._chunk_parts(fill_ghosts)))
        n_edge_chunks = sum(1 for w in order if parts[w][1])
        fill = None

        def zinv(win):
            self._win(win)
            if il:
                self.call("mcpm_slab_zinv3_il", _p(dst), self._interior_il(f3_ext))
            else:
                for c in range(3):
                    self.call("mcpm_slab_zinv", self._spec(dst, c), self._interior(f3_ext[c]), self.Me, 1)

        cw = self.nxl // C
        for k, w in enumerate(order):
            yield
            hA[w].wait()
            self._win((w * cw, cw))
            self.call("mcpm_slab_ycol2", _p(src), _p(dst), 1, 1, 0, 1)                # A -> force spectrum 0
            hG[w].wait()
            self.call("mcpm_slab_ycol2", _p(src), _p(dst), 1, 1, 0, 2)                # G -> force spectra 1, 2
            inner, edges = parts[w]
            for e in edges:                                                          # the edge planes first ...
                zinv(e)
            if edges:
                n_edge_chunks -= 1
                if n_edge_chunks == 0 and fill_ghosts:                               # ... so that the ghost fill runs under the rest
                    fill = self.halo_fill_x(f3_ext, async_op=True) if il else self.halo_fill(f3_ext, async_op=True)
            if inner is not None:
                zinv(inner)
        self._win((0, self.nxl))
        if fill is not None:
            yield
            fill.wait()

    def force_meshes_vjp(self, *args, **kw):
        _exhaust(self.force_meshes_vjp_gen(*args, **kw))

    def force_meshes_vjp_gen(self, fbar3_ext, rhobar_ext, ghost_adds=None, fill_ghosts=False):
        """GENERATOR (see force_meshes_gen).  fbar3_ext: three cotangent meshes (3, nxe, ny, nz) (ghosts added, or
        `ghost_adds[c]` handles still in flight).  Writes the interior of rhobar_ext (and its ghosts if fill_ghosts)."""
        ss, C = self.ss, self.chunks
        adds = [a for a in (ghost_adds or []) if a is not None]
        parts = self._chunk_parts(bool(adds))
        ha, hb, aliased = [None] * C, [None] * C, False

        def zy(win):      # three z transforms, then a = FFTy(f_bar_x) and b = ky FFTy(f_bar_y) + kz FFTy(f_bar_z)
            self._win(win)
            for c in range(3):
                self.call("mcpm_slab_zfwd", self._interior(fbar3_ext, c), self.Me, self._spec(self.s3a, c), 1)
            self.call("mcpm_slab_ycol2", _p(self.s3a), _p(self.s3b), 0, 0, 1, 3)

        def send(w):
            nonlocal aliased
            ha[w], aliased = self._a2a_chunk(self.s6a[3 * ss:4 * ss], self.s3b[:ss], w)
            hb[w], _ = self._a2a_chunk(self.s6a[4 * ss:5 * ss], self.s3b[ss:2 * ss], w)

        for w, (inner, edges) in enumerate(parts):
            if inner is not None:
                zy(inner)
            if not edges:
                send(w)
        if adds:
            if not all(getattr(a, "done", False) for a in adds):
                yield
            for a in adds:
                a.wait()
        for w, (inner, edges) in enumerate(parts):
            for e in edges:
                zy(e)
            if edges:
                send(w)
        self._win((0, self.nxl))
        yield
        for h in ha + hb:
            h.wait()
        # (a, b) received side by side in s6a[3 ss : 5 ss], or still in s3b[0 : 2 ss] when the all-to-all was aliased
        x_in = self.s3b if aliased else self.s6a[3 * ss:5 * ss]
        self.call("mcpm_slab_xfused", _p(x_in), _p(self.s1a), 1)
        order = self._edge_first()
        h1 = {}
        for w in order:
            h1[w], aliased = self._a2a_chunk(self.s1b, self.s1a, w)
        y_in, y_out = (self.s1a, self.s1b) if aliased else (self.s1b, self.s1a)
        parts = dict(enumerate(self._chunk_parts(fill_ghosts)))
        n_edge_chunks = sum(1 for w in order if parts[w][1])
        fill = None
        cw = self.nxl // C

        def zinv(win):
            self._win(win)
            self.call("mcpm_slab_zinv", _p(y_out), self._interior(rhobar_ext), self.Me, 1)

        for w in order:
            yield
            h1[w].wait()
            self._win((w * cw, cw))
            self.call("mcpm_slab_ycol", _p(y_in), _p(y_out), 1, +1, 1, 0)
            inner, edges = parts[w]
            for e in edges:
                zinv(e)
            if edges:
                n_edge_chunks -= 1
                if n_edge_chunks == 0 and fill_ghosts:
                    fill = self.halo_fill_x(rhobar_ext, async_op=True)
            if inner is not None:
                zinv(inner)
        self._win((0, self.nxl))
        if fill is not None:
            yield
            fill.wait()

    # ---- lpt on slabs (nbody.py:634-667 at the lattice, read_order = 1) -------------------------------------
    def spec_to_meshes(self, spec_full, out_ext, nc):
        """Replicated plain half-spectrum -> nc = 3 force meshes or nc = 6 Hessian meshes (interiors of out_ext[c])."""
        ss = self.ss
        self.call("mcpm_slab_xfused", _p(spec_full), _p(self.s6a), 2 if nc == 3 else 3)
        a2a = [self._a2a_all(self.s6b[c * ss:(c + 1) * ss], self.s6a[c * ss:(c + 1) * ss]) for c in range(nc)]
        for c in range(nc):
            a2a[c].wait()
            self.call("mcpm_slab_ycol", self._spec(self.s6b, c), self._spec(self.s6a, c), 1, +1, 1, 0)
            self.call("mcpm_slab_zinv", self._spec(self.s6a, c), self._interior(out_ext, c), self.Me, 1)

    def meshes_to_spec_bar(self, meshes_ext, spec_bar_full, nc):
        """Adjoint of spec_to_meshes: writes (nc = 3) or accumulates (nc = 6) this rank's y rows of spec_bar_full."""
        ss = self.ss
        a2a = []
        for c in range(nc):
            self.call("mcpm_slab_zfwd", self._interior(meshes_ext, c), self.Me, self._spec(self.s6a, c), 1)
            self.call("mcpm_slab_ycol", self._spec(self.s6a, c), self._spec(self.s6b, c), 1, -1, 0, 1)
            a2a.append(self._a2a_all(self.s6a[c * ss:(c + 1) * ss], self.s6b[c * ss:(c + 1) * ss]))
        for h in a2a:
            h.wait()
        self.call("mcpm_slab_xfused", _p(self.s6a), _p(spec_bar_full), 4 if nc == 3 else 5)

    def _h6(self):
        if self.h6 is None:
            self.h6 = torch.zeros((6, self.nxe) + self.shape[1:], dtype=torch.float32, device=self.device)
        return self.h6

    def lpt(self, spec, lpt_order, g, g2, dg2dg, dpos, vel):
        """dpos, vel (Nl,3) of this rank's particles from the replicated half-spectrum `spec`."""
        self.spec_to_meshes(spec, self.f3, 3)
        self.call("mcpm_lpt_accum_f32", _p(self.f3), float(g), 1.0, 1, _p(dpos), _p(vel))
        if lpt_order == 2:
            h6 = self._h6()
            self.spec_to_meshes(spec, h6, 6)
            self.call("mcpm_hessian_combine_f32", _p(h6), _p(self.rho))
            self.force_meshes(self.rho, self.f3, fill_ghosts=False)
            self.call("mcpm_lpt_accum_f32", _p(self.f3), -float(g2), -float(dg2dg), 0, _p(dpos), _p(vel))

    def lpt_vjp(self, spec, lpt_order, g, g2, dg2dg, xb, vb):
        """Cotangents (xb, vb) of this rank's (dpos, vel) -> (init_mesh_bar, [g_bar, g2_bar, dg2dg_bar]), both
        already summed over ranks (init_mesh_bar is the full half-spectrum on every rank)."""
        out = torch.zeros(tuple(spec.shape), dtype=torch.complex64, device=spec.device)
        sbar = torch.zeros(3, dtype=torch.float64, device=spec.device)
        self.spec_to_meshes(spec, self.f3, 3)
        self.call("mcpm_lattice_dot_f32", _p(self.f3), _p(xb), None, _p(sbar))
        self.call("mcpm_lattice_scatter_f32", _p(xb), _p(vb), float(g), 1.0, _p(self.f3))
        self.meshes_to_spec_bar(self.f3, out, 3)
        if lpt_order == 2:
            h6 = self._h6()
            self.spec_to_meshes(spec, h6, 6)
            self.call("mcpm_hessian_combine_f32", _p(h6), _p(self.rho))
            self.force_meshes(self.rho, self.f3, fill_ghosts=False)
            self.call("mcpm_lattice_dot_f32", _p(self.f3), _p(xb), _p(vb), C.c_void_p(sbar.data_ptr() + 8))
            self.call("mcpm_lattice_scatter_f32", _p(xb), _p(vb), -float(g2), -float(dg2dg), _p(self.f3))
            self.force_meshes_vjp(self.f3, self.rho)
            self.call("mcpm_hessian_combine_vjp_f32", _p(h6), _p(self.rho), _p(h6))
            self.meshes_to_spec_bar(h6, out, 6)
        out = self.comm.all_reduce_sum(out)
        sb = self.comm.all_reduce_sum(sbar).cpu().numpy()
        return out, np.array([sb[0], -sb[1], -sb[2]])

    # ---- one BullFrog step and its adjoint -----------------------------------------------------------------
    def reset_depth(self):
        """Forget the displacement history (call before the first step of a new trajectory: its depth is then the full G)."""
        self._dhist, self._dpending, self._dused = [], None, []

    def _harvest_depth(self):
        """Collects the measurement enqueued by the previous `set_depth` (its device work was enqueued a whole step ago, so
        the wait is normally free) and checks that the depth that step USED covered what its particles needed."""
        if getattr(self, "_dpending", None) is None:
            return
        host, ev, order = self._dpending
        self._dpending = None
        if host == "native":            # taken by the library at the end of the step before last (ev = its sequence number)
            val, ok = C.c_float(), C.c_int()
            self.call("mcpm_slab_dmax_read", int(ev), C.byref(val), C.byref(ok))
            if not ok.value:
                raise RuntimeError("ghost-depth measurement lost (more than four native steps without set_depth)")
            dmax = float(val.value)
        else:
            if ev is not None:
                ev.synchronize()
            dmax = float(host[0])
        self._dhist.append(dmax)
        need = min(self.G, int(math.floor(dmax)) + 1 + (1 if order > 2 else 0))
        if self._dused and self._dused[-1] < need:
            raise RuntimeError(f"ghost depth {self._dused[-1]} was predicted for a step whose particles reach {need} planes "
                               f"beyond the slab (max |d_x| = {dmax:.2f}): the result of that step is wrong; use "
                               f"adaptive_ghost=False or restart the trajectory with reset_depth()")

    def set_depth(self, x, paint_order=2, depth=None):
        """Ghost depth of a step whose particles sit at displacements x: a CIC deposit / gather reaches at most
        floor(max |d_x|) + 1 planes beyond the slab (order-3/4 stencils one more), so only those planes (plus spares) are
        exchanged instead of all G.  The host is NOT stopped to learn max |d_x| of this step: the maximum over ranks is
        enqueued (tiny all-reduce + pinned copy) and read one step later, and this step's depth is PREDICTED from the
        previous steps' maxima (last value + twice its last increase + one plane).  Every rank predicts from the same
        all-reduced numbers, so the depths agree.  The prediction is verified when the measurement arrives
        (`_harvest_depth` raises if it was too small); the first step of a trajectory uses the full depth.
        adaptive_ghost=False keeps the full depth always."""
        if not self.adaptive_ghost:
            self.ge = self.G
            return
        if depth is not None:       # the adjoint of a step revisits the forward step's positions: reuse its depth
            self.ge = int(depth)
            return
        if not hasattr(self, "_dhist"):
            self.reset_depth()
        self._harvest_depth()
        tok, self._dmax_token = self._dmax_token, None
        if (self.native and tok is not None and isinstance(tok[0], str) and tok[1].data_ptr() == x.data_ptr()
                and tok[1].shape == x.shape and tok[2] == x._version and tok[3] == self._kd_gen):
            # x came out of the latest native step (the token keeps that tensor alive, so the address cannot have been reused,
            # and torch has not written to it since: version counter), which enqueued max |d_x| over ranks itself under the
            # sequence number the token carries (no collective from here)
            self._dpending = ("native", tok[4], paint_order)
            self._predict_depth(paint_order)
            return
        if tok is not None and not isinstance(tok[0], torch.Tensor):
            tok = None
        if (tok is not None and tok[0].data_ptr() == x.data_ptr() and tok[0].shape == x.shape and tok[1] == x._version
                and tok[2] == self._kd_gen):
            # these positions came out of this plan's latest kick_drift and torch has not written to them since (an in-place
            # update bumps the version counter; the token keeps the tensor alive, so the address cannot belong to another one;
            # any other kick_drift on this plan drops the token): max |d_x| is already in the slots
            dmx = self.dmax.max().reshape(1).view(torch.float32)
        else:
            dmx = x[:, 0].abs().max()
        host, ev = self.comm.all_reduce_max_async(dmx)
        self._dpending = (host, ev, paint_order)
        self._predict_depth(paint_order)

    def _predict_depth(self, paint_order):
        h = self._dhist
        if not h:
            self.ge = self.G
        else:
            growth = max(h[-1] - h[-2], 0.0) if len(h) > 1 else 0.5 * h[-1]
            pred = h[-1] + 2.0 * growth + 1.0
            reach = int(math.floor(pred)) + 1 + (1 if paint_order > 2 else 0)
            self.ge = max(1, min(self.G, reach + 1))
        self._dused.append(self.ge)

    def finish_depth(self):
        """Verifies the last step's predicted depth (stops the host once; call at the end of a forward trajectory)."""
        if self.adaptive_ghost:
            self._harvest_depth()

    def step(self, *args, **kw):
        _exhaust(self.step_gen(*args, **kw))

    def step_vjp(self, *args, **kw):
        _exhaust(self.step_vjp_gen(*args, **kw))

    def step_gen(self, x, v, alpha, beta, tau, f3_out, x_out, v_out, paint_order=2):
        """GENERATOR (see force_meshes_gen; `step` runs it to the end).  x, v: (Nl,3) local state; f3_out: (nxe, ny, nz, 3) receives the ghost-filled INTERLEAVED force mesh (ghost planes
        beyond the exchanged depth `self.ge` are left as they were: no particle of this step reads them)."""
        self.set_depth(x, paint_order)
        if self.native:
            seq = int(lib.mcpm_slab_dmax_seq(self.h))
            self._kd_gen += 1
            self._dmax_token = None
            self._native_call("mcpm_slab_step_f32", _p(x), _p(v), float(alpha), float(beta), float(tau), int(paint_order), int(self.ge),
                              _p(f3_out), _p(x_out), _p(v_out))
            # this step's measurement (of x_out) is number `seq` of the library's ring; a step whose input misses the token
            # simply never reads its predecessor's entry (the ring is overwritten after four)
            self._dmax_token = ("native", x_out, x_out._version, self._kd_gen, seq)
            return
        self.call("mcpm_paint_f32", _p(x), self.Nl, POS_LATTICE, None, 1, 1.0, paint_order, _p(self.rho), 0)
        yield from self.force_meshes_gen(self.rho, f3_out, rho_add=self.halo_add_x(self.rho, async_op=True), il=True)
        self.call("mcpm_kick_drift_il_f32", _p(x), _p(v), self.Nl, POS_LATTICE, _p(f3_out), paint_order, float(alpha),
                  float(beta), float(tau), _p(x_out), _p(v_out))
        self._dmax_token = (x_out, x_out._version, self._kd_gen)

    def step_vjp_gen(self, x, v, f3, alpha, beta, tau, xb, vb, abar_ptr, bbar_ptr, dtau_ddg=1.0, dgbar_ptr=None, paint_order=2,
                     depth=None, next_beta_tau=None):
        """GENERATOR (`step_vjp` runs it to the end).  Adjoint of `step` (f3: its interleaved force mesh): xb, vb (cotangents of its outputs) are updated in place.  `depth`: the ghost depth the
        forward step used (`self.ge` after `step`), saving its re-measurement.  `next_beta_tau`: (beta, tau) of the step
        whose adjoint comes next (the previous step of the sweep): its force cotangent is then written by this call's
        particle kernel and picked up by the next call instead of a separate pass."""
        self.set_depth(x, paint_order, depth)
        if self.native:
            nb, nt = next_beta_tau if next_beta_tau is not None else (0.0, 0.0)
            self._native_call("mcpm_slab_step_vjp_f32", _p(x), _p(v), _p(f3), float(alpha), float(beta), float(tau), int(paint_order),
                              int(self.ge), _p(xb), _p(vb), abar_ptr, bbar_ptr, float(dtau_ddg), dgbar_ptr,
                              1 if next_beta_tau is not None else 0, float(nb), float(nt))
            return
        fb = C.c_void_p()
        self.call("mcpm_plan_chained_fb", float(beta), float(tau), _p(xb), _p(vb), C.byref(fb))
        if not fb.value:
            self.call("mcpm_kick_f32", _p(vb), _p(xb), self.Nl, float(beta), float(beta * tau), _p(self.Fb))
            fb = _p(self.Fb)
        self.call("mcpm_paint3_f32", _p(x), self.Nl, POS_LATTICE, fb, paint_order, _p(self.f3), 0)
        add = self.halo_add(self.f3, async_op=True)      # ONE exchange for the three components, under the inner planes' passes
        yield from self.force_meshes_vjp_gen(self.f3, self.rho, [add, add, add], fill_ghosts=True)
        if next_beta_tau is not None:
            self.call("mcpm_plan_hint_next_adjoint", float(next_beta_tau[0]), float(next_beta_tau[1]))
        self.call("mcpm_step_adjoint_particles_il_f32", _p(x), _p(v), _p(f3), _p(self.rho), float(alpha), float(beta),
                  float(tau), paint_order, _p(xb), _p(vb), abar_ptr, bbar_ptr, float(dtau_ddg), dgbar_ptr)


class SlabCtx:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def nbody_bf_slab(cosmo, init_mesh, a0=0., a1=1., n_steps=5, paint_order=2, lpt_order=2, comm=None, ghost=16,
                  integrator="bullfrog", return_ctx=False, slab=None, native=None, chunks=None):
    """Slab-decomposed `nbody_bf` (nbody.py:967-1002).  `init_mesh` is the full half-spectrum, replicated on every
    rank.  Returns this rank's (disp, vel), each (N/P, 3): displacement from the lattice and velocity of the
    particles whose lattice plane lies in the rank's slab (rows [r N/P, (r+1) N/P) of the global arrays)."""
    spec = nbody._c64(init_mesh)
    shape = nbody.ch2rshape(spec.shape)
    pm = slab if slab is not None else SlabPM(shape, comm, ghost, native=native, chunks=chunks)
    n_steps = int(n_steps)
    dg, alphas, betas, lpt_s = nbody._step_scalars(cosmo, a0, a1, n_steps, integrator)
    K = n_steps
    states = torch.empty((K + 1, 2, pm.Nl, 3), dtype=torch.float32, device=spec.device)
    f3s = torch.zeros((K, pm.nxe) + shape[1:] + (3,), dtype=torch.float32, device=spec.device) if return_ctx else None
    pm.lpt(spec, int(lpt_order), lpt_s[0], lpt_s[1], lpt_s[2], states[0, 0], states[0, 1])      # slab-decomposed LPT start
    states[0, 0] += states[0, 1] * (dg / 2)
    depths = []
    pm.reset_depth()
    for i in range(K):
        tau = dg / 2 if i == K - 1 else dg
        pm.step(states[i, 0], states[i, 1], alphas[i], betas[i], tau, f3s[i] if return_ctx else pm.force_mesh_il(),
                states[i + 1, 0], states[i + 1, 1], paint_order)
        depths.append(pm.ge)
    pm.finish_depth()
    out = (states[K, 0], states[K, 1])
    if return_ctx:
        return out, SlabCtx(pm=pm, spec=spec, states=states, f3s=f3s, dg=dg, alphas=alphas, betas=betas, lpt_s=lpt_s,
                            n_steps=K, lpt_order=int(lpt_order), paint_order=int(paint_order), depths=depths)
    return out


def nbody_bf_slab_vjp(ctx, disp_bar, vel_bar):
    """Reverse sweep.  disp_bar, vel_bar: this rank's (N/P, 3) cotangents.  Returns (init_mesh_bar, scalar_bars) with
    the full init_mesh_bar on every rank (real-pair convention) and the scalar cotangents summed over ranks."""
    pm, K = ctx.pm, ctx.n_steps
    dev = ctx.spec.device
    xb = nbody._f32(disp_bar, (pm.Nl, 3)).clone()
    vb = nbody._f32(vel_bar, (pm.Nl, 3)).clone()
    sbar = torch.zeros(2 * K + 1, dtype=torch.float64, device=dev)
    for i in reversed(range(K)):
        tau = ctx.dg / 2 if i == K - 1 else ctx.dg
        pm.step_vjp(ctx.states[i, 0], ctx.states[i, 1], ctx.f3s[i], ctx.alphas[i], ctx.betas[i], tau, xb, vb,
                    C.c_void_p(sbar.data_ptr() + 8 * i), C.c_void_p(sbar.data_ptr() + 8 * (K + i)),
                    0.5 if i == K - 1 else 1.0, C.c_void_p(sbar.data_ptr() + 8 * 2 * K), ctx.paint_order, depth=ctx.depths[i],
                    next_beta_tau=(ctx.betas[i - 1], ctx.dg) if i > 0 else None)
    sbar[2 * K] += 0.5 * (xb.double() * ctx.states[0, 1].double()).sum()    # initial half drift x'_0 = x_0 + v_0 dg/2
    vb += xb * (ctx.dg / 2)
    sbar = pm.comm.all_reduce_sum(sbar).cpu().numpy()
    out, ls = pm.lpt_vjp(ctx.spec, ctx.lpt_order, ctx.lpt_s[0], ctx.lpt_s[1], ctx.lpt_s[2], xb, vb)
    return out, {"alpha": sbar[:K].copy(), "beta": sbar[K:2 * K].copy(), "g": ls[0], "g2": ls[1], "dg2dg": ls[2],
                 "dg": float(sbar[2 * K])}
