"""Worker bodies for the multi-process slab tests (spawned with torch.multiprocessing)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _init(rank, world, port, backend="gloo"):
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        import torch
        td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        td.init_process_group(backend, rank=rank, world_size=world)
    return td


def cpu_comm_worker(rank, world, port, out_dir):
    """gloo / CPU: communicator primitives and the ghost-plane algebra against a serial numpy model."""
    import torch
    td = _init(rank, world, port)
    from montecosmo_amd.dist import TorchComm, HaloMixin
    comm = TorchComm()
    nx, ny, nz, G = 8 * world, 4, 6, 3
    nxl = nx // world
    rng = np.random.default_rng(0)                      # same global field on every rank

    class Halo(HaloMixin):
        pass

    h = Halo()
    h.comm, h.G, h.nxl = comm, G, nxl
    # halo_fill: ghosts must equal the periodic neighbours' planes of the global mesh
    glob = rng.standard_normal((2, nx, ny, nz)).astype(np.float32)
    ext = torch.zeros((2, nxl + 2 * G, ny, nz))
    ext[:, G:G + nxl] = torch.from_numpy(glob[:, rank * nxl:(rank + 1) * nxl])
    h.halo_fill(ext)
    idx = np.arange(rank * nxl - G, (rank + 1) * nxl + G) % nx
    assert np.array_equal(ext.numpy(), glob[:, idx]), "halo_fill"
    # the same through the un-staged branch (what RCCL runs: point-to-point batch_isend_irecv of the contiguous plane runs,
    # dist.TorchComm.exchange), synchronous and asynchronous
    h2 = Halo()
    h2.comm, h2.G, h2.nxl = TorchComm(stage=False), G, nxl
    for async_op in (False, True):
        ext2 = torch.zeros((2, nxl + 2 * G, ny, nz))
        ext2[:, G:G + nxl] = torch.from_numpy(glob[:, rank * nxl:(rank + 1) * nxl])
        h2.halo_fill(ext2, async_op=async_op).wait()
        assert np.array_equal(ext2.numpy(), glob[:, idx]), "halo_fill, un-staged point-to-point"
    # halo_add: every rank paints an extended slab; the sum over ranks of the unfolded slabs is the global mesh
    ext_all = rng.standard_normal((world, nxl + 2 * G, ny, nz)).astype(np.float32)
    want = np.zeros((nx, ny, nz), np.float64)
    for r in range(world):
        ii = np.arange(r * nxl - G, (r + 1) * nxl + G) % nx
        np.add.at(want, ii, ext_all[r].astype(np.float64))
    mine = torch.from_numpy(ext_all[rank].copy())
    h.halo_add(mine)
    got = mine[G:G + nxl].numpy()
    assert np.allclose(got, want[rank * nxl:(rank + 1) * nxl], atol=1e-5), "halo_add"
    mine2 = torch.from_numpy(ext_all[rank].copy())
    h2.halo_add(mine2, async_op=True).wait()
    assert np.allclose(mine2[G:G + nxl].numpy(), want[rank * nxl:(rank + 1) * nxl], atol=1e-5), "halo_add, split all-to-all"
    # all_to_all in the packed transpose layout: [dest][xl][yl][k] -> [x][yl][k]
    nyl, nk = 2, 3
    full = rng.standard_normal((nx, nyl * world, nk)).astype(np.float32)          # global [x][y][k]
    mine = full[rank * nxl:(rank + 1) * nxl]                                     # my x planes, all y
    send = np.stack([mine[:, d * nyl:(d + 1) * nyl] for d in range(world)])       # [d][xl][yl][k]
    recv = torch.empty(send.shape)
    comm.all_to_all(recv, torch.from_numpy(send.copy()))
    assert np.array_equal(recv.numpy().reshape(nx, nyl, nk), full[:, rank * nyl:(rank + 1) * nyl]), "all_to_all transpose"
    # complex payloads travel as float pairs
    cs = torch.from_numpy((send + 1j * send).astype(np.complex64).reshape(-1).copy())
    cr = torch.empty_like(cs)
    comm.all_to_all(cr, cs)
    assert np.array_equal(cr.numpy().real.reshape(nx, nyl, nk), full[:, rank * nyl:(rank + 1) * nyl])
    t = comm.all_reduce_sum(torch.tensor([float(rank + 1)], dtype=torch.float64))
    assert float(t) == world * (world + 1) / 2
    g = comm.all_gather_cat(torch.full((2, 3), float(rank)))
    assert g.shape == (2 * world, 3) and float(g[2 * rank, 0]) == rank
    td.barrier()
    td.destroy_process_group()
    open(os.path.join(out_dir, f"ok_{rank}"), "w").write("ok")


def gpu_slab_worker(rank, world, port, out_dir, n, n_steps, backend="gloo"):
    """Several ranks sharing cuda:0 (gloo, staged through the host), or one rank through RCCL itself ("nccl": self
    send/recv and a one-rank all-to-all, un-staged and asynchronous as on a multi-GPU node): the slab path against
    the single-GPU path."""
    import torch
    torch.cuda.set_device(0)
    td = _init(rank, world, port, backend)
    from montecosmo_amd import nbody, bricks, synth, dist
    shape = (n, n, n)
    spec = synth.init_mesh(n, seed=3, rms_disp=1.5)
    cosmo = bricks.Planck18()
    comm = dist.TorchComm()
    (d, v), ctx = dist.nbody_bf_slab(cosmo, spec, a0=0.1, a1=1.0, n_steps=n_steps, comm=comm, ghost=8, return_ctx=True)
    rng = np.random.default_rng(5)
    xb = rng.standard_normal((n ** 3, 3)).astype(np.float32)
    vb = rng.standard_normal((n ** 3, 3)).astype(np.float32)
    Nl = n ** 3 // world
    mb, sb = dist.nbody_bf_slab_vjp(ctx, xb[rank * Nl:(rank + 1) * Nl], vb[rank * Nl:(rank + 1) * Nl])
    d_all, v_all = comm.all_gather_cat(d), comm.all_gather_cat(v)
    if rank == 0:
        (lp, v1), c1 = nbody.nbody_bf(cosmo, spec, nbody.LatticePos.regular(shape), a0=0.1, a1=1.0, n_steps=n_steps,
                                      return_ctx=True, lattice_out=True)
        mb1, sb1 = nbody.nbody_bf_vjp(c1, xb, vb)

        def rel(a, b):
            a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
            return float(np.linalg.norm(a - b) / np.linalg.norm(b))

        res = {"disp": rel(d_all, lp.disp), "vel": rel(v_all, v1), "grad": rel(mb, mb1),
               "alpha": float(np.abs(sb["alpha"] - sb1["alpha"]).max() / np.abs(sb1["alpha"]).max()),
               "beta": float(np.abs(sb["beta"] - sb1["beta"]).max() / np.abs(sb1["beta"]).max()),
               "lpt_scalars": float(max(abs(sb[k] - sb1[k]) for k in ("g", "g2", "dg2dg")) / abs(sb1["g"]))}
        import json
        json.dump(res, open(os.path.join(out_dir, "result.json"), "w"))
    td.barrier()
    td.destroy_process_group()


def gpu_slab_golden_worker(rank, world, port, out_dir, name, backend="gloo"):
    """The slab path with `world` ranks against float64 ORACLE vectors in the layout of tests/golden/nbody_*.npz (final
    state, final density, gradient and scalar cotangents of BASELINE config 4's workload at test size), not against the
    single-GPU HIP path.  `name`: a file under tests/golden/ or an absolute path (the slab FFT needs axes >= 64, so the
    test writes 64^3 oracle vectors to a temporary file)."""
    import json
    import torch
    torch.cuda.set_device(0)
    td = _init(rank, world, port, backend)
    from montecosmo_amd import nbody, bricks, dist
    g = np.load(name if os.path.isabs(name) else os.path.join(ROOT, "tests", "golden", name))
    n, n_steps, a0 = int(g["n"]), int(g["n_steps"]), float(g["a0"])
    shape = (n, n, n)
    comm = dist.TorchComm()
    (d, v), ctx = dist.nbody_bf_slab(bricks.Planck18(), g["init_mesh"], a0=a0, a1=1.0, n_steps=n_steps, comm=comm, ghost=8,
                                     return_ctx=True)
    Nl = n ** 3 // world
    mb, sb = dist.nbody_bf_slab_vjp(ctx, g["pos_bar"][rank * Nl:(rank + 1) * Nl], g["vel_bar"][rank * Nl:(rank + 1) * Nl])
    d_all, v_all = comm.all_gather_cat(d), comm.all_gather_cat(v)
    oob = ctx.pm.out_of_ghost()
    if rank == 0:
        def rel(a, b):
            a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
            dt = np.complex128 if np.iscomplexobj(b) else np.float64
            return float(np.linalg.norm(a.astype(dt) - b.astype(dt)) / np.linalg.norm(b.astype(dt)))
        lp = nbody.LatticePos(d_all, shape)
        res = {"disp": rel(d_all, g["final_disp"]), "vel": rel(v_all, g["final_vel"]),
               "density": rel(nbody.paint(lp, shape), g["final_density"]),
               "cell_mismatch": float(np.mean(np.any(nbody.cell_index(lp, shape).cpu().numpy() != g["final_cell"], axis=1))),
               "grad": rel(mb, g["init_mesh_bar"]),
               "alpha": float(np.abs(sb["alpha"] - g["alpha_bar"]).max() / np.abs(g["alpha_bar"]).max()),
               "beta": float(np.abs(sb["beta"] - g["beta_bar"]).max() / np.abs(g["beta_bar"]).max()),
               "lpt_scalars": float(np.abs(np.array([sb["g"], sb["g2"], sb["dg2dg"]]) - g["lpt_scalar_bars"]).max()
                                    / np.abs(g["lpt_scalar_bars"]).max()),
               "oob": int(oob)}
        json.dump(res, open(os.path.join(out_dir, "result.json"), "w"))
    td.barrier()
    td.destroy_process_group()


def gpu_interleaved_worker(rank, world, port, out_dir, n, n_steps, backend="gloo"):
    """bench.run_interleaved: two independent trajectories (own streams, own process groups) issued alternately must give,
    bit for bit, what each gives when it runs alone (every kernel of the path is order-independent)."""
    import json
    import torch
    td = _init(rank, world, port, backend) if world > 1 or backend == "nccl" else None
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    a = bench.SlabRunner(n, n_steps, dev, 8, True, seed=0)
    grp = td.new_group(backend=td.get_backend()) if td is not None else None
    with torch.cuda.stream(torch.cuda.Stream(dev)):
        b = bench.SlabRunner(n, n_steps, dev, 8, True, seed=1, group=grp)
    torch.cuda.synchronize()
    solo = []
    for r in (a, b):
        with torch.cuda.stream(r.stream):
            r.run(n_steps)
        torch.cuda.synchronize()
        solo.append((r.states.pair(n_steps), r.xb.clone(), r.vb.clone(), r.sbar.clone()))
        r.states.zero_from(1)
        r.sbar.zero_()
    bench.run_interleaved([a, b], n_steps)
    torch.cuda.synchronize()
    ok = True
    for r, (st, xb, vb, sb) in zip((a, b), solo):
        ok = ok and torch.equal(r.states.pair(n_steps), st) and torch.equal(r.xb, xb) and torch.equal(r.vb, vb)
        ok = ok and bool(torch.isfinite(xb).all()) and float(xb.abs().max()) > 0
        # the scalar cotangents accumulate over calls (+=): after zeroing they must come back the same
        ok = ok and torch.allclose(r.sbar, sb, rtol=1e-12, atol=0)
    differ = not torch.equal(a.states.pair(n_steps), b.states.pair(n_steps))       # two different trajectories, really
    if td is not None:
        t = torch.tensor([float(ok and differ)])
        td.all_reduce(t, op=td.ReduceOp.MIN)
        ok_all = bool(t.item() > 0)
        td.barrier()
        td.destroy_process_group()
    else:
        ok_all = ok and differ
    if rank == 0:
        json.dump({"ok": ok_all}, open(os.path.join(out_dir, "result.json"), "w"))


def gpu_native_equal_worker(rank, world, port, out_dir, n, n_steps, backend="gloo", chunks=None):
    """csrc/slab.hip (one library call per step, the library issues the exchanges) against the Python-issued path of
    dist.SlabPM: same kernels, same windows, same order -> every output must be BITWISE equal.  world > 1: gloo ranks sharing
    cuda:0, the library's exchanges go through the host-callback transport (dist.HostStagedOps); backend "nccl" with one rank:
    the plan-owned RCCL communicator against torch's."""
    import json
    import torch
    torch.cuda.set_device(0)
    td = _init(rank, world, port, backend) if (world > 1 or backend == "nccl") else None
    from montecosmo_amd import bricks, dist, synth
    shape = (n, n, n)
    spec = synth.init_mesh(n, seed=3, rms_disp=2.0)
    rng = np.random.default_rng(5)
    Nl = n ** 3 // world
    xb = rng.standard_normal((n ** 3, 3)).astype(np.float32)[rank * Nl:(rank + 1) * Nl]
    vb = rng.standard_normal((n ** 3, 3)).astype(np.float32)[rank * Nl:(rank + 1) * Nl]
    outs = []
    for native in (False, True):
        comm = dist.TorchComm() if td is not None else dist.LocalComm()
        (d, v), ctx = dist.nbody_bf_slab(bricks.Planck18(), spec, a0=0.1, a1=1.0, n_steps=n_steps, comm=comm, ghost=8,
                                         return_ctx=True, native=native, chunks=chunks)
        assert ctx.pm.native == native
        mb, sb = dist.nbody_bf_slab_vjp(ctx, xb, vb)
        outs.append((d.clone(), v.clone(), mb.clone(), sb, list(ctx.depths), ctx.pm.out_of_ghost()))
        del ctx
    a, b = outs
    ok = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    ok = ok and all(np.array_equal(np.asarray(a[3][k]), np.asarray(b[3][k])) for k in a[3])
    ok = ok and a[4] == b[4] and a[5] == b[5] == 0
    ok = ok and bool(torch.isfinite(a[2].real).all()) and float(a[0].abs().max()) > 0.5
    if td is not None:
        t = torch.tensor([float(ok)])
        if backend == "nccl":
            t = t.cuda()
        td.all_reduce(t, op=td.ReduceOp.MIN)
        ok = bool(t.item() > 0)
        td.barrier()
        td.destroy_process_group()
    if rank == 0:
        json.dump({"ok": ok, "depths": a[4], "depths_native": b[4]}, open(os.path.join(out_dir, "result.json"), "w"))


def gpu_native_fallback_worker(rank, world, port, out_dir, n, n_steps):
    """The library's transport fails its self-test on ONE rank (mcpm_slab_comm_selftest patched to return MCPM_E_RCCL there):
    every rank must give it up together and keep issuing the exchanges through torch.distributed, with results bitwise those
    of a run that asked for that path; asking for the native path explicitly must raise instead."""
    import json
    import torch
    torch.cuda.set_device(0)
    td = _init(rank, world, port, "gloo")
    from montecosmo_amd import bricks, dist, synth
    real = dist.lib.mcpm_slab_comm_selftest
    calls = []

    class Patched:
        def __getattr__(self, name):
            if name == "mcpm_slab_comm_selftest":
                def selftest(h):
                    calls.append(1)
                    rc = real(h)                       # the real one runs (it is collective) and passes ...
                    return -5 if rank == world - 1 else rc     # ... but the last rank reports MCPM_E_RCCL
                return selftest
            return getattr(real_lib, name)

    real_lib, dist.lib = dist.lib, Patched()
    spec = synth.init_mesh(n, seed=3, rms_disp=2.0)
    outs = []
    try:
        for native in (False, None):
            (d, v), ctx = dist.nbody_bf_slab(bricks.Planck18(), spec, a0=0.1, a1=1.0, n_steps=n_steps, comm=dist.TorchComm(), ghost=8,
                                             return_ctx=True, native=native)
            outs.append((d.clone(), v.clone(), ctx.pm.native, ctx.pm.native_fallback))
            del ctx
        raised = False
        try:
            dist.SlabPM((n, n, n), dist.TorchComm(), 8, native=True)
        except Exception:
            raised = True
    finally:
        dist.lib = real_lib
    a, b = outs
    ok = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] is False and b[2] is False and a[3] is None and b[3] is not None
    ok = ok and raised and len(calls) == 2 and float(a[0].abs().max()) > 0.5
    t = torch.tensor([float(ok)])
    td.all_reduce(t, op=td.ReduceOp.MIN)
    td.barrier()
    td.destroy_process_group()
    if rank == 0:
        json.dump({"ok": bool(t.item() > 0), "reason": b[3]}, open(os.path.join(out_dir, "result.json"), "w"))
