#!/usr/bin/env python
"""Benchmark of the PM hot path on MI355X (contract: see DESIGN.md "Measurement").

Metric (BASELINE.json): PM forward+adjoint steps per second.  One "step" = one BullFrog drift-kick-drift
step (CIC paint -> R2C -> k-space -> 3 C2R -> fused read+kick+drift) plus its hand-written adjoint, fp32,
N = n^3 particles on an n^3 mesh, synthetic inputs of SURVEY.md 8(d).  The timed region runs K forward steps
(writing checkpoints) followed by their K adjoint steps, with every input resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mesh n]

N > 1 runs one rank per GPU (RCCL): the SAME mesh is x-slab decomposed over the ranks (montecosmo_amd/dist.py: ghost
planes point-to-point, one all-to-all per FFT transpose), so the total work is fixed and scaling is "strong".  The ranks
come from `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (WORLD_SIZE / RANK / LOCAL_RANK in
the environment), or, when WORLD_SIZE is unset, `python bench.py --gpus N` starts the N rank processes itself (fresh
children, before this process touches the GPU) and relays rank 0's JSON line.  `--gpus` must equal the world size; the
line reports the world size the communicator observed (`rccl_world`).  `--replicas` instead runs N independent copies
(the reference's own multi-device mode: independent chains, script.py:13-20; "weak", no collective on the data path).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
B_PER_CELL_CYCLE = 100.0         # paint -> Poisson -> read force cycle
B_PER_CELL_STEP = 492.0          # forward + adjoint DKD step
B_PER_CELL_FWD = 124.0           # forward DKD step alone: paint 16 + R2C 8 + k-space 16 + three C2R 24 + fused read/kick/drift 60 (48 N + 12 M)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed forward+adjoint steps (any K: the n-step simulation is replayed; 50 = five replays, 0.6 s at 512^3)")
    ap.add_argument("--n-steps", type=int, default=10, help="BullFrog steps of the simulated a0 -> a1 run (the workload; memory ~ n-steps)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mesh", type=int, default=int(os.environ.get("MCPM_BENCH_MESH", "512")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-record", action="store_true", help="skip the second (--cpu-mesh) GPU record and the CPU baseline on it: "
                    "profiler passes want one mesh size per run, kernel names do not carry it")
    ap.add_argument("--replicas", action="store_true", help="N > 1: independent replicas instead of slabs")
    ap.add_argument("--force-slab", action="store_true", help="N = 1: run the slab code path with a local-copy communicator "
                    "(measures the slab path's own overhead: ghost planes, windowed passes, host calls)")
    ap.add_argument("--forward-only", action="store_true", help="time forward steps only (BASELINE config 2)")
    ap.add_argument("--ghost", type=int, default=8)
    ap.add_argument("--chains", type=int, default=0, help="slab path: ALSO time this many independent trajectories issued alternately "
                    "on their own streams / process groups (extra `interleaved_chains` record; `value` stays the single trajectory). "
                    "0 = 2 when N > 1, else off")
    ap.add_argument("--fixed-ghost", action="store_true", help="slabs: always exchange all ghost planes (default: only the "
                    "planes each step's displacements can reach)")
    ap.add_argument("--cpu-mesh", type=int, default=256, help="mesh of the second GPU record and of the CPU baseline (unscaled)")
    ap.add_argument("--rehearse", action="store_true", help="launch path only: start / join the ranks, rendezvous over gloo on the "
                    "CPU, barrier + max-over-ranks, print the JSON skeleton; touches no GPU (CPU test of --gpus N)")
    return ap.parse_args()


class _States:
    """states[i, c]: checkpoint i, c = 0 position offsets x'_i, 1 velocities v_i -- (N, 3) views into the runner's flat buffer."""

    def __init__(self, views):
        self.views = views

    def __getitem__(self, key):
        i, c = key
        return self.views[i][c]

    def pair(self, i):
        """(2, N, 3) copy of checkpoint i."""
        return torch.stack([self.views[i][0], self.views[i][1]])

    def zero_from(self, i0):
        for x, v in self.views[i0:]:
            x.zero_()
            v.zero_()


class Runner:
    """K forward steps + K adjoint steps through the step-level C ABI, on pre-allocated HBM buffers.

    Layout of the particle arrays (DESIGN finding 29).  The step kernels stream four to seven (N, 3) arrays at the same particle
    index at the same time.  Laid out back to back they are 12 N bytes apart -- 1.5 GiB at 512^3 --, i.e. IN PHASE in every
    low address bit, and whether the streams then meet in the same memory channel is decided by the upper physical bits: the
    per-process placement that made the adjoint particle kernel bimodal (2.45 or 2.75 ms).  The placement logic lives in the
    LIBRARY since round 4: the runner allocates one flat buffer, hands it to `mcpm_plan_probe_particle_pitch` (which times the
    adjoint particle kernel on it for three candidate pitches and keeps the fastest) and lays its arrays out at the pitch that
    comes back; `MCPM_PARTICLE_PITCH` (floats) fixes it.  The choice is reported (`layout` in the JSON line)."""

    forward_only = False

    def __init__(self, n, K, device, forward_only=False):
        from montecosmo_amd import nbody, bricks, synth
        from montecosmo_amd._lib import lib
        self.forward_only = bool(forward_only)
        self.nbody = nbody
        self.n, self.K = n, K
        self.device = device
        shape = (n, n, n)
        self.plan = nbody.get_plan(shape)
        N, M = self.plan.N, self.plan.M
        self.N, self.M = N, M
        cosmo = bricks.Planck18()
        self.dg, self.alphas, self.betas, self.lpt_s = nbody._step_scalars(cosmo, 0.0, 1.0, K, "bullfrog")
        spec = synth.init_mesh(n, seed=0, rms_disp=2.0)
        self.spec = torch.from_numpy(spec).to(device)
        f32 = dict(dtype=torch.float32, device=device)
        self.fmesh = torch.empty((K, 3, n, n, n), **f32)          # force meshes per step
        self.sbar = torch.zeros((2 * K + 1,), dtype=torch.float64, device=device)
        # one flat buffer for the 2 (K + 1) checkpoint arrays, the two loss cotangents and the two running cotangents, sized for the
        # largest pitch the library may choose (3 N + 17472 floats)
        narr = 2 * (K + 1) + 4
        flat = torch.empty(narr * (3 * N + 17472), dtype=torch.float32, device=device)
        pitch = C.c_int64()
        env = os.environ.get("MCPM_PARTICLE_PITCH")
        if env is not None:
            self.plan.call("mcpm_plan_set_particle_pitch", int(env))
            self.plan.call("mcpm_plan_particle_pitch", C.byref(pitch))
            source = "MCPM_PARTICLE_PITCH"
        elif self.forward_only:
            self.plan.call("mcpm_plan_particle_pitch", C.byref(pitch))
            source = "default (forward only: nothing to probe)"
        else:
            self.plan.call("mcpm_plan_probe_particle_pitch", self.p(flat), flat.numel(), C.byref(pitch))
            source = "mcpm_plan_probe_particle_pitch"
        pitch = int(pitch.value)
        self.layout = {"pitch_floats": pitch, "shift_bytes": 4 * (pitch - 3 * N), "source": source}
        arr = lambda j: flat[j * pitch: j * pitch + 3 * N].view(N, 3)
        j = 2 * (K + 1)
        self._flat = flat
        self.states = _States([(arr(2 * i), arr(2 * i + 1)) for i in range(K + 1)])       # (x'_i, v_i) checkpoints
        self.pos_bar, self.vel_bar, self.xb, self.vb = arr(j), arr(j + 1), arr(j + 2), arr(j + 3)
        rng = np.random.default_rng(1)
        self.pos_bar.copy_(torch.from_numpy(rng.standard_normal((N, 3), dtype=np.float32)))
        self.vel_bar.copy_(torch.from_numpy(rng.standard_normal((N, 3), dtype=np.float32)))
        self.init_state()

    def p(self, t):
        return C.c_void_p(t.data_ptr())

    def init_state(self):
        """LPT initial conditions + first half drift (untimed set-up of the step loop)."""
        x, v = self.states[0, 0], self.states[0, 1]
        self.plan.call("mcpm_lpt_f32", self.p(self.spec), 2, float(self.lpt_s[0]), float(self.lpt_s[1]), float(self.lpt_s[2]),
                       0, 0, self.p(x), self.p(v))
        self.plan.call("mcpm_drift_f32", self.p(x), self.p(v), self.N, float(self.dg / 2), self.p(x))

    def forward(self, steps):
        K = self.K
        for i in range(steps):
            tau = self.dg / 2 if i == K - 1 else self.dg
            self.plan.call("mcpm_bullfrog_step_f32", self.p(self.states[i, 0]), self.p(self.states[i, 1]),
                           float(self.alphas[i]), float(self.betas[i]), float(tau), 2, self.p(self.fmesh[i]),
                           self.p(self.states[i + 1, 0]), self.p(self.states[i + 1, 1]))

    def backward(self, steps):
        K = self.K
        for i in reversed(range(steps)):
            tau = self.dg / 2 if i == K - 1 else self.dg
            if i > 0:   # chain: this step's particle kernel also writes the force cotangent of step i-1
                self.plan.call("mcpm_plan_hint_next_adjoint", float(self.betas[i - 1]), float(self.dg))
            # the first reverse step reads the loss cotangents where they are and writes the running ones (no copy of 24 N bytes
            # per trajectory); the others update the running ones in place
            first = i == steps - 1
            self.plan.call("mcpm_bullfrog_step_vjp_from_f32", self.p(self.states[i, 0]), self.p(self.states[i, 1]),
                           self.p(self.fmesh[i]), float(self.alphas[i]), float(self.betas[i]), float(tau), 2,
                           self.p(self.pos_bar if first else self.xb), self.p(self.vel_bar if first else self.vb),
                           self.p(self.xb), self.p(self.vb), C.c_void_p(self.sbar.data_ptr() + 8 * i),
                           C.c_void_p(self.sbar.data_ptr() + 8 * (K + i)), 0.5 if i == K - 1 else 1.0,
                           C.c_void_p(self.sbar.data_ptr() + 8 * 2 * K))

    def run(self, steps):
        """`steps` forward(+adjoint) steps: the K-step simulation replayed in blocks (a last, shorter block covers the rest:
        its forward steps 0..r-1 and their adjoints)."""
        while steps > 0:
            s = min(steps, self.K)
            self.forward(s)
            if not self.forward_only:
                self.backward(s)
            steps -= s

    def force_cycle_ms(self, reps=10):
        """pm_forces(pos, mesh_shape) on its own (paint -> Poisson -> 3-component read; nbody.py:583-604) on the
        evolved particles of the last checkpoint: HIP events on the plan's stream, mean of `reps` calls."""
        forces = self.xb                                   # (N, 3) scratch, overwritten by the next backward()
        x = self.states[self.K, 0]
        args = (self.p(x), self.N, 1, 2, 0, 0, 0, 0.0, self.p(forces))   # MCPM_POS_LATTICE, CIC, fd = inf, no kcut
        self.plan.call("mcpm_pm_forces_f32", *args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            self.plan.call("mcpm_pm_forces_f32", *args)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    def profile(self):
        """Per-stage HIP-event timings of one more pass over the same K steps (events on the plan's stream)."""
        from montecosmo_amd._lib import lib
        self.plan.call("mcpm_plan_profile", 1)
        self.forward(self.K)
        fwd = self._read_profile()
        if not self.forward_only:
            self.backward(self.K)
        bwd = self._read_profile()
        self.plan.call("mcpm_plan_profile", 0)
        names = [lib.mcpm_stage_name(i).decode() for i in range(len(fwd[0]))]
        return names, fwd, bwd

    def _read_profile(self):
        from montecosmo_amd._lib import lib
        nmax = 16
        ms = (C.c_double * nmax)()
        by = (C.c_double * nmax)()
        calls = (C.c_int64 * nmax)()
        ns = lib.mcpm_plan_profile_read(self.plan.h, nmax, ms, by, calls)
        assert ns > 0
        return list(ms)[:ns], list(by)[:ns], list(calls)[:ns]


class SlabRunner:
    """The same K forward + K adjoint steps on an x-slab of the mesh (one rank of N)."""

    def __init__(self, n, K, device, ghost, adaptive_ghost=True, seed=0, group=None):
        """Everything (plan, buffers, kernels, collectives) lives on the torch stream that is current at construction
        (`self.stream`), so that a second, independent trajectory built under another stream can run concurrently."""
        from montecosmo_amd import nbody, bricks, synth, dist
        self.n, self.K = n, K
        shape = (n, n, n)
        import torch.distributed as td
        self.stream = torch.cuda.current_stream(device)
        self.comm = dist.TorchComm(group) if (td.is_available() and td.is_initialized()) else dist.LocalComm()
        self.pm = dist.SlabPM(shape, self.comm, ghost, device, adaptive_ghost=adaptive_ghost)
        pm = self.pm
        cosmo = bricks.Planck18()
        self.dg, self.alphas, self.betas, self.lpt_s = nbody._step_scalars(cosmo, 0.0, 1.0, K, "bullfrog")
        spec = torch.from_numpy(synth.init_mesh(n, seed=seed, rms_disp=2.0)).to(device)
        f32 = dict(dtype=torch.float32, device=device)
        # particle arrays: one flat buffer at a fixed pitch of 3 N + 1088 floats (4 KB + 256 B), so that the streams of a step
        # kernel are not in phase in every low address bit (Runner's doc string; no per-process probe here: the ranks would have
        # to agree on it)
        shift = int(os.environ.get("MCPM_PARTICLE_SHIFT_FLOATS", "1088"))
        assert shift % 4 == 0
        pitch = 3 * pm.Nl + shift
        self._flat = torch.empty((2 * (K + 1) + 4) * pitch, **f32)
        arr = lambda j: self._flat[j * pitch: j * pitch + 3 * pm.Nl].view(pm.Nl, 3)
        self.states = _States([(arr(2 * i), arr(2 * i + 1)) for i in range(K + 1)])
        self.layout = {"pitch_floats": pitch, "shift_bytes": 4 * shift, "source": "fixed (slab ranks must agree)"}
        self.f3s = torch.zeros((K, pm.nxe, n, n, 3), **f32)          # interleaved force meshes per step
        # slab-decomposed LPT start (untimed set-up) + first half drift
        pm.lpt(spec, 2, self.lpt_s[0], self.lpt_s[1], self.lpt_s[2], self.states[0, 0], self.states[0, 1])
        x0 = self.states[0, 0]
        x0 += self.states[0, 1] * (self.dg / 2)
        del spec
        torch.cuda.empty_cache()
        rng = np.random.default_rng(1 + pm.rank + 1000 * seed)
        j = 2 * (K + 1)
        self.pos_bar, self.vel_bar, self.xb, self.vb = arr(j), arr(j + 1), arr(j + 2), arr(j + 3)
        self.pos_bar.copy_(torch.from_numpy(rng.standard_normal((pm.Nl, 3), dtype=np.float32)))
        self.vel_bar.copy_(torch.from_numpy(rng.standard_normal((pm.Nl, 3), dtype=np.float32)))
        self.sbar = torch.zeros((2 * K + 1,), dtype=torch.float64, device=device)
        self.depths = [None] * K

    # one trajectory = fwd_begin, fwd_step(0..s-1), fwd_end, bwd_begin, bwd_step(s-1..0): split so that two trajectories can
    # be issued alternately (run_interleaved)
    def fwd_begin(self):
        self.pm.reset_depth()           # a new trajectory: first step at full ghost depth, then predicted depths

    def fwd_step(self, i):
        K = self.K
        tau = self.dg / 2 if i == K - 1 else self.dg
        self.pm.step(self.states[i, 0], self.states[i, 1], self.alphas[i], self.betas[i], tau, self.f3s[i],
                     self.states[i + 1, 0], self.states[i + 1, 1])
        self.depths[i] = self.pm.ge

    def fwd_end(self):
        self.pm.finish_depth()          # verifies the last prediction (the only host stop of the trajectory)

    def bwd_begin(self):
        self.xb.copy_(self.pos_bar)
        self.vb.copy_(self.vel_bar)

    def bwd_step(self, i):
        K = self.K
        tau = self.dg / 2 if i == K - 1 else self.dg
        self.pm.step_vjp(self.states[i, 0], self.states[i, 1], self.f3s[i], self.alphas[i], self.betas[i], tau,
                         self.xb, self.vb, C.c_void_p(self.sbar.data_ptr() + 8 * i),
                         C.c_void_p(self.sbar.data_ptr() + 8 * (K + i)), 0.5 if i == K - 1 else 1.0,
                         C.c_void_p(self.sbar.data_ptr() + 8 * 2 * K), depth=self.depths[i],
                         next_beta_tau=(self.betas[i - 1], self.dg) if i > 0 else None)

    def traj_gen(self, s):
        """One trajectory of s forward + s adjoint steps as a generator of compute segments (see run_interleaved)."""
        K = self.K
        self.fwd_begin()
        for i in range(s):
            tau = self.dg / 2 if i == K - 1 else self.dg
            yield from self.pm.step_gen(self.states[i, 0], self.states[i, 1], self.alphas[i], self.betas[i], tau, self.f3s[i],
                                        self.states[i + 1, 0], self.states[i + 1, 1])
            self.depths[i] = self.pm.ge
        self.fwd_end()
        self.bwd_begin()
        for i in reversed(range(s)):
            tau = self.dg / 2 if i == K - 1 else self.dg
            yield from self.pm.step_vjp_gen(self.states[i, 0], self.states[i, 1], self.f3s[i], self.alphas[i], self.betas[i], tau,
                                            self.xb, self.vb, C.c_void_p(self.sbar.data_ptr() + 8 * i),
                                            C.c_void_p(self.sbar.data_ptr() + 8 * (K + i)), 0.5 if i == K - 1 else 1.0,
                                            C.c_void_p(self.sbar.data_ptr() + 8 * 2 * K), depth=self.depths[i],
                                            next_beta_tau=(self.betas[i - 1], self.dg) if i > 0 else None)

    def forward(self, steps):
        self.fwd_begin()
        for i in range(steps):
            self.fwd_step(i)
        self.fwd_end()

    def backward(self, steps):
        self.bwd_begin()
        for i in reversed(range(steps)):
            self.bwd_step(i)

    def run(self, steps):
        while steps > 0:
            s = min(steps, self.K)
            self.forward(s)
            self.backward(s)
            steps -= s

    def profile(self):
        from montecosmo_amd._lib import lib
        self.pm.call("mcpm_plan_profile", 1)
        self.forward(self.K)
        fwd = self._read_profile()
        self.backward(self.K)
        bwd = self._read_profile()
        self.pm.call("mcpm_plan_profile", 0)
        names = [lib.mcpm_stage_name(i).decode() for i in range(len(fwd[0]))]
        return names, fwd, bwd

    def _read_profile(self):
        from montecosmo_amd._lib import lib
        nmax = 16
        ms, by, calls = (C.c_double * nmax)(), (C.c_double * nmax)(), (C.c_int64 * nmax)()
        ns = lib.mcpm_plan_profile_read(self.pm.h, nmax, ms, by, calls)
        assert ns > 0
        return list(ms)[:ns], list(by)[:ns], list(calls)[:ns]


def run_interleaved(runners, steps, offset=None, token=None):
    """`steps` forward+adjoint steps of EACH of the independent trajectories in `runners` (SlabRunner s built under their own
    torch streams and process groups) as a software pipeline: a trajectory is a generator that yields whenever it has just
    launched a collective and would next wait for it (dist.SlabPM.step_gen); the driver then issues the next compute
    segment of ANOTHER trajectory, so that one's all-to-alls / ghost exchanges run under the other's kernels -- the
    multi-chain form in which a sampler uses the slab path (the reference's multi-device mode is independent chains,
    script.py:13-20).
      offset: trajectory c runs `offset` segments behind trajectory c-1, so that the communication-heavy Poisson section of
              one meets the particle kernels of the other (segments per forward step: kick-drift + paint + inner z/y passes |
              edge z/y | fused x | y of A | y of G + z | ...);
      token : compute segments of different trajectories are chained by events (never concurrent): what overlaps a
              collective is then exactly what was issued between its launch and its wait.  Off by default: it also makes a
              trajectory wait for the other's collectives (measured slower on the one-rank RCCL proxy, tools/chain_sweep.sh)."""
    offset = int(os.environ.get("MCPM_CHAIN_OFFSET", "2")) if offset is None else offset
    token = (os.environ.get("MCPM_CHAIN_TOKEN", "0") == "1") if token is None else token
    K, n = runners[0].K, len(runners)
    while steps > 0:
        s = min(steps, K)
        gens = [r.traj_gen(s) for r in runners]
        live, adv, tok = [True] * n, [0] * n, None
        while any(live):
            for c, r in enumerate(runners):
                if not live[c] or (c > 0 and live[c - 1] and adv[c - 1] - adv[c] < offset):
                    continue
                with torch.cuda.stream(r.stream):
                    if token and tok is not None:
                        r.stream.wait_event(tok)
                    try:
                        next(gens[c])
                    except StopIteration:
                        live[c] = False
                    if token:
                        tok = torch.cuda.Event()
                        tok.record(r.stream)
                adv[c] += 1
        steps -= s


def pmc_traffic(stage, n):
    """HBM-side bytes per launch of `stage`, from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes committed under
    profiles/ (separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 and as calibrated here
    on axpy_kernel, whose byte count is known).  bench.py cannot collect PMC counters itself; null when the file
    has no entry for this mesh.  Returns (bytes, file used): the newest profiles/r*_pmc_traffic.json
    that holds the stage; the line names the file (`traffic_source`) so that a stale table is visible."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))      # rNN_...: the newest round sorts last
    for path in reversed(files):
        try:
            d = json.load(open(path))
            return d[str(n)][stage]["bytes_per_launch"], os.path.basename(path)
        except Exception:
            continue
    return None, None


def host_threads():
    """CPU threads this process may use (the GPU box gives one GPU's share of the host, not all of its cores)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 64))


def cpu_baseline(runner, n_sample_steps=5):
    """The reference-equivalent CPU path (the float64 oracle with its threaded back end: scipy.fft on all allowed host
    threads + the OpenMP paint / read kernels of oracle/csrc/pm_kernels.c; the JAX reference itself cannot run here), timed
    on forward+adjoint DKD steps of the SAME trajectory the GPU just ran at runner.n^3: the GPU's checkpoint of step i is
    handed to the oracle, which runs the step and its adjoint in float64.  Not scaled: value is steps/s at runner.n^3."""
    from oracle import pm_oracle as o, background as obg
    n, K, N = runner.n, runner.K, runner.N
    shape = (n, n, n)
    threads = host_threads()
    prev = o.set_threads(threads)
    try:
        cos = obg.Planck18()
        pos = o.regular_pos(shape)
        rng = np.random.default_rng(1)
        xb, vb = rng.standard_normal((N, 3)), rng.standard_normal((N, 3))
        dg = float(runner.dg)
        g0 = float(o.a2g(cos, 0.0))
        steps = list(range(max(0, (K - n_sample_steps) // 2), min(max(0, (K - n_sample_steps) // 2) + n_sample_steps, K)))   # ~12 s at 256^3 on 64 threads
        states = [(runner.states[i, 0].double().cpu().numpy(), runner.states[i, 1].double().cpu().numpy()) for i in steps]
        t0 = time.perf_counter()
        for i, (xh, v) in zip(steps, states):
            t = g0 + i * dg
            o.dkd_vjp(pos + xh - v * (dg / 2), v, xb, vb, dg, float(o.alpha_bf(cos, t, dg)), t + dg / 2, shape)
        dt = time.perf_counter() - t0
    finally:
        o.set_threads(prev)
    return {"value": round(len(steps) / dt, 4), "unit": "steps/s", "cores": threads, "kind": "port", "mesh": n,
            "sample": f"{len(steps)} forward+adjoint DKD steps (steps {steps[0]}..{steps[-1]} of {K}) of the {n}^3 bench trajectory, "
                      f"float64 oracle with scipy.fft workers + OpenMP paint/read on {threads} threads, took {dt:.1f} s "
                      f"(host reports {os.cpu_count()} cores); not scaled"}


def sub_record(n, NS, K, W, device):
    """The same measurement at a second mesh (the metric names 256^3 and 512^3): timed region + pm_forces."""
    r = Runner(n, NS, device)
    r.run(W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.run(K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pmf = r.force_cycle_ms()
    M = float(n) ** 3
    return r, {"mesh": n, "value": round(K / dt, 3), "unit": "steps/s", "steps": K, "ms_per_step": round(dt / K * 1e3, 4),
               "pm_forces_ms": round(pmf, 4), "pm_forces_frac_of_hbm_peak": round(B_PER_CELL_CYCLE * M / (pmf * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "fwd_adj_step_frac_of_hbm_peak": round(B_PER_CELL_STEP * M / (dt / K) / 1e9 / HBM_PEAK_GBS, 4)}


def spawn_ranks(nproc):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this process has not touched the GPU
    and never will), one per GPU, rendezvous on 127.0.0.1, and relay their output.  Returns the worst exit code."""
    import socket
    import subprocess
    # the rendezvous port stays bound (SO_REUSEADDR) until the ranks have been started, so that no other process is handed it
    sk = socket.socket()
    sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    procs = []
    for r in range(nproc):
        env = dict(os.environ, WORLD_SIZE=str(nproc), RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(nproc),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    sk.close()
    # poll all ranks: the first failure ends the others (they would sit in a collective for ever), and so does the deadline
    deadline = time.time() + float(os.environ.get("MCPM_BENCH_DEADLINE", "1500"))
    rc, live = 0, list(procs)
    while live:
        for pr in list(live):
            r = pr.poll()
            if r is not None:
                live.remove(pr)
                rc = max(rc, abs(r))
        if live and (rc != 0 or time.time() > deadline):
            if rc == 0:
                rc = 5
                print("bench.py: ranks still running at the deadline; ending them", file=sys.stderr, flush=True)
            for pr in live:
                pr.terminate()
            t_end = time.time() + 10
            for pr in live:
                try:
                    pr.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    pr.kill()
            break
        time.sleep(0.05)
    return rc


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch with --nproc-per-node {args.gpus}, "
                 f"or leave WORLD_SIZE unset and let bench.py start the ranks")
    if args.rehearse:
        import torch.distributed as td
        if world > 1:
            td.init_process_group(backend="gloo")
            td.barrier()
            t = torch.tensor([float(rank)], dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            seen, top = td.get_world_size(), float(t.item())
            td.barrier()
            td.destroy_process_group()
        else:
            seen, top = 1, 0.0
        if rank == 0:
            print(json.dumps({"rehearsal": True, "n_gpus": world, "rccl_world": seen, "comm_backend": "gloo" if world > 1 else None,
                              "max_rank_seen": top, "steps": args.steps, "warmup": args.warmup}))
        return
    # MCPM_BENCH_DIST=1 with one rank initialises the process group anyway: on a one-GPU box it drives the slab path
    # through the real RCCL calls (self send/recv, one-rank all-to-all) instead of the local-copy communicator
    dist = world > 1 or os.environ.get("MCPM_BENCH_DIST") == "1"
    if dist:
        # The slab path overlaps collectives (RCCL's stream) with kernels (the plan's stream, and a second trajectory's).  HIP
        # multiplexes streams onto 4 hardware queues by default, and streams that share a queue run in submission order:
        # the trace of the one-rank RCCL run showed the plan's stream, the second trajectory's and a communicator's on ONE
        # queue (16.15 -> 15.45 ms per step with 8 queues, profiles/r02_chain_pipeline.txt).  Read at HIP initialisation,
        # which has not happened yet (importing torch does not initialise the GPU).
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    ndev = max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank % ndev)
    device = torch.device("cuda", local_rank % ndev)
    if dist:
        import torch.distributed as td
        # RCCL ("nccl") on a real multi-GPU node; MCPM_BENCH_BACKEND=gloo only rehearses the script with several
        # ranks sharing one GPU (collectives staged through the host) and is never a measurement
        backend = os.environ.get("MCPM_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            td.init_process_group(backend="nccl", device_id=device)
        else:
            td.init_process_group(backend=backend)
    K, W, n, NS = args.steps, args.warmup, args.mesh, args.n_steps

    slab = (dist and not args.replicas) or args.force_slab
    assert not (args.forward_only and slab), "--forward-only is a single-GPU configuration"
    r = SlabRunner(n, NS, device, args.ghost, not args.fixed_ghost) if slab else Runner(n, NS, device, forward_only=args.forward_only)
    w = W
    if w > 0:                          # W untimed warm-up steps (code objects, caches, allocator)
        r.run(w)

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    r.run(K)
    barrier()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=device if td.get_backend() == "nccl" else "cpu")
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    prof = r.profile() if (slab or rank == 0) else None     # slabbed: collective, every rank takes part
    pmf_ms = r.force_cycle_ms() if (rank == 0 and not slab) else None
    if rank == 0:
        M = float(n) ** 3          # whole mesh: stage times below are rank 0's, which holds 1/world of it when slabbed
        steps_per_s = (1 if slab else world) * K / dt
        names, fwd, bwd = prof
        stages = {}
        for i, nm in enumerate(names):
            ms = fwd[0][i] + bwd[0][i]
            calls = fwd[2][i] + bwd[2][i]
            by = fwd[1][i] + bwd[1][i]
            if calls:
                stages[nm] = {"ms_total": round(ms, 3), "launches": calls, "ms_per_launch": round(ms / calls, 4),
                              "algorithmic_GBps": round(by / (ms * 1e-3) / 1e9, 1)}
        dom = max(stages, key=lambda k: stages[k]["ms_total"])
        fwd_ms = sum(fwd[0])
        bwd_ms = sum(bwd[0])
        # force cycle = paint + R2C + k-space + 3 C2R + read(+kick+drift) of the forward pass
        Mloc = M / world if slab else M    # cells whose stages rank 0 timed
        cyc_ms = sum(fwd[0][names.index(k)] for k in ("paint", "fft_r2c", "kspace", "fft_c2r", "kick_drift")) / NS   # the profile pass runs the NS-step block once
        step_ms = (fwd_ms + bwd_ms) / NS
        step_b = B_PER_CELL_FWD if args.forward_only else B_PER_CELL_STEP
        out = {
            "metric": "PM forward steps/sec" if args.forward_only else "PM forward+adjoint steps/sec", "value": round(steps_per_s, 3), "unit": "steps/s",
            "n_gpus": world, "rccl_world": (td.get_world_size() if dist else 1), "comm_backend": (td.get_backend() if dist else None),
            "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 3),
            "higher_is_better": True, "scaling": "weak" if args.replicas else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n}^3 mesh, {n}^3 particles, {NS}-step BullFrog {'forward only' if args.forward_only else 'forward+VJP'}, CIC, 2LPT start (untimed), "
                                   f"rms displacement 2 cells; " + ("single GPU" if world == 1 else
                                                                    (f"x-slab decomposed over {world} GPUs (ghost {args.ghost} planes, RCCL all-to-all FFT transpose)"
                                                                     if slab else f"{world} independent replicas")),
                       "mesh": n, "n_steps": NS, "parallelism": (("slab1 (RCCL, one rank)" if dist else "slab1 (local-copy communicator)") if slab else "single") if world == 1 else (f"slab{world}" if slab else f"replicas{world}")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": stages[dom]["algorithmic_GBps"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(stages[dom]["algorithmic_GBps"] / HBM_PEAK_GBS, 4),
                         "traffic": pmc_traffic(dom, n)[0] if world == 1 else None,
                         "traffic_source": pmc_traffic(dom, n)[1] if world == 1 else None},
            # "ms": the stages of the cycle inside a step, where the read also kicks and drifts (60 B/particle instead of
            # the 36 the 100 B/cell figure counts); "pm_forces_*": the function pm_forces itself, which is what 100 B/cell describes
            "force_cycle": {"ms": round(cyc_ms, 4), "algorithmic_GBps": round(B_PER_CELL_CYCLE * Mloc / (cyc_ms * 1e-3) / 1e9, 1),
                            "frac_of_hbm_peak": round(B_PER_CELL_CYCLE * Mloc / (cyc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            # (a forward-only run is priced with the forward step's own bytes, under its own key)
            ("fwd_step" if args.forward_only else "fwd_adj_step"): {
                "ms_events": round(step_ms, 4), "bytes_per_cell": step_b,
                "algorithmic_GBps": round(step_b * Mloc / (step_ms * 1e-3) / 1e9, 1),
                "frac_of_hbm_peak": round(step_b * Mloc / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "stages": stages,
        }
        if pmf_ms is not None:
            out["force_cycle"].update({"pm_forces_ms": round(pmf_ms, 4),
                                       "pm_forces_GBps": round(B_PER_CELL_CYCLE * M / (pmf_ms * 1e-3) / 1e9, 1),
                                       "pm_forces_frac_of_hbm_peak": round(B_PER_CELL_CYCLE * M / (pmf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        if True:
            out["layout"] = dict(r.layout, note="particle arrays in one flat buffer, array j at j * pitch_floats; the pitch is chosen by the library "
                                             "(mcpm_plan_probe_particle_pitch: adjoint particle kernel timed on this buffer for 3 candidate pitches, untimed set-up)")
        if slab:
            out["comm_and_host_ms_per_step"] = round(dt / K * 1e3 - step_ms, 3)   # wall minus rank-0 kernel stages
            out["deposits_beyond_ghost_rank0"] = r.pm.out_of_ghost()                # must be 0
            out["ghost_planes_exchanged_per_step"] = [int(d) for d in r.depths if d is not None]   # of --ghost allocated
            pm = r.pm       # who issues the exchanges: the library (csrc/slab.hip) unless its transport failed its self-test
            out["slab_transport"] = (("library, local copies" if world == 1 and not dist else "library, plan-owned RCCL communicator (self-test passed)")
                                     if pm.native else f"torch.distributed ({pm.native_fallback or 'MCPM_SLAB_NATIVE=0'})")
            out["a2a_chunks"] = r.pm.chunks      # all-to-alls per transposed spectrum (dist.SlabPM chunks; MCPM_SLAB_CHUNKS)
        if world == 1 and not slab and not args.forward_only and not args.no_sub_record:
            # the metric names 256^3 as well: a second, smaller record in the same line, and the CPU baseline on ITS trajectory
            if n != args.cpu_mesh:
                r2, out[f"mesh_{args.cpu_mesh}"] = sub_record(args.cpu_mesh, NS, K, W, device)
            else:
                r2 = r
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(r2)
                # SURVEY 8(d) lists 64^3 and 128^3 as well: the same measurement on their own trajectories (a second or two each)
                del r2
                small = {}
                for m in (128, 64):
                    if m in (n, args.cpu_mesh):
                        continue
                    rs, rec = sub_record(m, NS, K, W, device)
                    small[str(m)] = dict(cpu_baseline(rs, n_sample_steps=NS), gpu_steps_per_s=rec["value"])
                    del rs
                out["cpu_baseline_small_meshes"] = small
    # ---- independent trajectories as a software pipeline (collective: every rank takes part).  Measured AFTER the line
    # above is complete, under a watchdog: whatever happens here (a second communicator that cannot be created, a hang), the
    # single-trajectory result is printed.
    nch = args.chains if args.chains > 0 else (2 if (slab and world > 1) else 1)
    if slab and nch > 1:
        import threading

        def bail():
            # a stuck collective cannot be left from Python: print what is finished and end THIS process with a failure code
            # (every rank runs its own watchdog), so that the launcher and spawn_ranks see the hang instead of rc 0
            if rank == 0:
                out["interleaved_chains"] = {"chains": nch, "error": "timed out"}
                print(json.dumps(out), flush=True)
            os._exit(3)

        timer = threading.Timer(float(os.environ.get("MCPM_CHAINS_TIMEOUT", "300")), bail)
        timer.daemon = True
        timer.start()
        try:
            runners = [r]
            for c in range(1, nch):
                grp = td.new_group(backend=td.get_backend()) if dist else None      # own communicator: no head-of-line blocking
                with torch.cuda.stream(torch.cuda.Stream(device)):
                    runners.append(SlabRunner(n, NS, device, args.ghost, not args.fixed_ghost, seed=c, group=grp))
            torch.cuda.synchronize()
            run_interleaved(runners, min(max(W, 1), NS))
            barrier()
            t0 = time.perf_counter()
            run_interleaved(runners, K)
            barrier()
            dtc = time.perf_counter() - t0
            if dist:
                t = torch.tensor([dtc], dtype=torch.float64, device=device if td.get_backend() == "nccl" else "cpu")
                td.all_reduce(t, op=td.ReduceOp.MAX)
                dtc = float(t.item())
            if rank == 0:
                out["interleaved_chains"] = {
                    "chains": nch, "offset": int(os.environ.get("MCPM_CHAIN_OFFSET", "2")), "value": round(nch * K / dtc, 3), "unit": "steps/s (all chains)",
                    "ms_per_step": round(dtc / (nch * K) * 1e3, 3), "vs_single_trajectory": round(nch * K / dtc / (K / dt), 3),
                    "note": f"{nch} independent {n}^3 trajectories (seeds 0..{nch - 1}) on their own streams and process groups, issued "
                            "as a software pipeline of compute segments (run_interleaved), so that one's all-to-all / ghost exchanges "
                            "run under the other's kernels; `value` above stays the single trajectory"}
            del runners
        except Exception as e:      # the other ranks may be stuck in a collective: their watchdogs end them
            if rank == 0:
                out["interleaved_chains"] = {"chains": nch, "error": repr(e)[:300]}
                print(json.dumps(out), flush=True)
            else:
                print(f"bench.py rank {rank}: interleaved chains failed: {e!r}", file=sys.stderr, flush=True)
            os._exit(4)
        finally:
            timer.cancel()
    if dist:
        td.barrier()
        td.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
