"""Builds profiles/rNN_pmc_traffic.json from two rocprofv3 PMC passes over bench.py (FETCH_SIZE, WRITE_SIZE).
usage: python tools/make_traffic_json.py <fetch_dir> <write_dir> <mesh> <out.json> <per_kernel.txt>
An existing <out.json> is updated (one entry per mesh), so the 512^3 and 256^3 passes share a file."""
import collections, csv, glob, json, os, sys

fetch_dir, write_dir, mesh, out_json, out_txt = sys.argv[1:6]
# kernel-name fragment -> stage of bench.py's stage table.  The small kernels of a tiled paint (prologue, coverage test,
# buckets, leftovers) serve both the density paint and the three-component paint: they are attributed in dispatch order to
# the paint whose tile kernel they surround (a prologue belongs to the tile kernel that follows it, the others to the one
# before them).
stage_of = [("paint3_tile_kernel", "paint3"), ("paint3_bucket_kernel", "paint3"), ("paint3_epilogue_kernel", "paint3"), ("paint_tile_kernel", "paint"),
            ("paint_bucket_kernel", "paint"), ("paint_epilogue_kernel", "paint"),
            ("paint_atomic_kernel", "paint"), ("paint_fxg_flush_kernel", "paint"),
            ("zfwd_kernel", "fft_r2c"), ("ycol2_kernel<512, false", "fft_r2c"), ("ycol2_kernel<256, false", "fft_r2c"),
            ("ycol2_kernel<512, true", "fft_c2r"), ("ycol2_kernel<256, true", "fft_c2r"), ("ycol_kernel<512, -1>", "fft_r2c"), ("ycol_kernel<256, -1>", "fft_r2c"),
            ("ycol_kernel<512, 1>", "fft_c2r"), ("ycol_kernel<256, 1>", "fft_c2r"), ("zinv_kernel", "fft_c2r"), ("zinv3_il_kernel", "fft_c2r"),
            ("xfused_kernel", "kspace"), ("kick_drift_kernel", "kick_drift"), ("step_adjoint_kernel", "step_adjoint"),
            ("axpby_kernel", "axpy"), ("axpy_kernel", "axpy")]
shared_before = ("tile_prologue_kernel", "halo_select_kernel", "halo_decide_kernel", "halo_tile_kernel", "box_tile_kernel")   # belong to the NEXT tile kernel
shared_after = ("coverage_duty_kernel", "paint_leftover_kernel", "absmax_kernel")   # belong to the PREVIOUS tile kernel
# the kernels that count as "one launch of the stage" (its main kernel); the others only add their bytes
main_kernels = ("paint_tile_kernel", "paint3_tile_kernel<", "paint_atomic_kernel", "zfwd_kernel", "ycol2_kernel", "ycol_kernel",
                "zinv_kernel", "zinv3_il_kernel", "xfused_kernel", "kick_drift_kernel", "step_adjoint_kernel", "axpby_kernel", "axpy_kernel")


def stage(name):
    for k, v in stage_of:
        if k in name:
            return v
    return None


tot = collections.defaultdict(lambda: {"fetch_kb": 0.0, "write_kb": 0.0, "launches": 0})
per_kernel = collections.defaultdict(lambda: {"fetch_kb": [], "write_kb": []})
for d, key in ((fetch_dir, "fetch_kb"), (write_dir, "write_kb")):
    f = max(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)      # the latest pass
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    current, pending = None, 0.0
    for r in rows:
        nm, v = r["Kernel_Name"], float(r["Counter_Value"])
        per_kernel[nm.split("(")[0]][key].append(v)
        if any(s in nm for s in shared_before):
            pending += v
            continue
        if any(s in nm for s in shared_after):
            if current:
                tot[current][key] += v
            continue
        st = stage(nm)
        if st in ("paint", "paint3") and "tile_kernel" in nm:
            # the f64 repaint launch (template arguments <F64 = true, 1024 threads, U, FAST = false>) follows its
            # fixed-point pass (<..., false, 512, 4, 0 | 1 | 2>): only the first pass opens a new paint
            if not ("paint3_tile_kernel" in nm and "<true, 1024," in nm):
                tot[st][key] += pending
                pending = 0.0
            current = st
        if st:
            tot[st][key] += v
            if key == "fetch_kb" and any(m in nm for m in main_kernels) and not ("paint3_tile_kernel" in nm and "<true, 1024," in nm):
                tot[st]["launches"] += 1      # (the f64 repaint launch of the three-component paint is not a paint of its own)
method = (
    "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --no-sub-record --no-cpu-baseline --warmup 0 "
    "[--mesh 256]`; counters are KB; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies "
    "128-B requests as 64 B), calibrated here on axpy_kernel / lpt_accum_kernel whose byte counts are known; the counters sit "
    "on the L2 fabric side, so Infinity-Cache hits are included. Per stage = sum over the stage's kernels (incl. the small "
    "prologue / coverage / bucket / leftover kernels of a tiled paint) / launches of the stage's main kernel.")
try:
    out = json.load(open(out_json))
except Exception:
    out = {}
out["_method"] = method
out[str(mesh)] = {}
for st, v in tot.items():
    n = max(v["launches"], 1)
    out[str(mesh)][st] = {"bytes_per_launch": round((2 * v["fetch_kb"] + v["write_kb"]) * 1024 / n),
                          "fetch_bytes_per_launch": round(2 * v["fetch_kb"] * 1024 / n),
                          "write_bytes_per_launch": round(v["write_kb"] * 1024 / n), "launches_profiled": n}
json.dump(out, open(out_json, "w"), indent=1)
with open(out_txt, "w") as fo:
    fo.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), per-kernel averages, raw KB (FETCH_SIZE not yet doubled)\n")
    for k, v in sorted(per_kernel.items()):
        if v["fetch_kb"]:
            fo.write(f"{k[:90]:92s} n={len(v['fetch_kb']):4d} FETCH_SIZE_avg_KB={sum(v['fetch_kb'])/len(v['fetch_kb']):14.1f} "
                     f"WRITE_SIZE_avg_KB={sum(v['write_kb'])/max(len(v['write_kb']),1):14.1f}\n")
for st, v in out[str(mesh)].items():
    print(st, {k: (f"{x/1e9:.2f} GB" if "bytes" in k else x) for k, x in v.items()})
