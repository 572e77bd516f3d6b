#!/usr/bin/env python3
"""HBM-side bytes per VQ row / per image from rocprofv3 PMC passes -> profiles/hbm_traffic_<workload>.json (read by bench.py).

    python tools/pmc_traffic.py WORKLOAD B BENCH_FETCH_DIR BENCH_WRITE_DIR VQ_FETCH_DIR VQ_WRITE_DIR VQ_ROWS OUT.json

Method (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE in SEPARATE passes, kernel-trace only; counters are
KiB per dispatch; FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of 16-B/lane streaming reads at 64 B --
calibrated on a 1 GiB copy in round 1, profiles/r01_vq_hbm_traffic.txt); WRITE_SIZE is taken as is.  Infinity-Cache hits
are counted, so these are fabric-side bytes (an upper bound on HBM bytes).  The VQ figure comes from a stream far beyond
the 256 MiB Infinity Cache (tools/vq_traffic.py), the conv figure from the bench workload itself."""
import collections, csv, glob, json, os, subprocess, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(d):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def resources(d):
    """VGPRs / accumulation VGPRs / scratch bytes per lane / LDS bytes of every kernel, from the kernel-trace CSV of a pass."""
    out = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"][:70]] = {"vgpr": int(r.get("VGPR_Count") or 0), "accum_vgpr": int(r.get("Accum_VGPR_Count") or 0),
                                          "scratch_bytes_per_lane": int(r.get("Scratch_Size") or r.get("Private_Segment_Size") or 0),
                                          "lds_bytes": int(r.get("LDS_Block_Size") or r.get("Group_Segment_Size") or 0)}
    return out


def main():
    wl, B = sys.argv[1], int(sys.argv[2])
    bf, bw, vf, vw = (per_kernel(p) for p in sys.argv[3:7])
    vq_rows, out = int(sys.argv[7]), sys.argv[8]
    is_vq = lambda k: "vq_" in k and any(t in k for t in ("kernel_d64", "vq_exact_kernel", "vq_stream", "vq_filter", "vq_cand", "vq_refine"))
    vqk = [k for k in vf if is_vq(k)]
    assert vqk, list(vf)
    k = max(vqk, key=lambda n: vf[n][0])
    vq_read, vq_write = 2 * vf[k][0] * 1024, vw[k][0] * 1024
    res = {"workload": wl, "per_gpu_batch": B,
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate kernel-trace passes; bytes = 2 x FETCH_SIZE KiB "
                     "(gfx950 correction for wide coalesced reads) + WRITE_SIZE KiB; fabric-side (Infinity-Cache hits counted)",
           "vq_kernel": k[:60], "vq_rows_measured": vq_rows,
           "vq_read_bytes_per_row": round(vq_read / vq_rows, 2), "vq_write_bytes_per_row": round(vq_write / vq_rows, 2),
           "vq_bytes_per_row": round((vq_read + vq_write) / vq_rows, 2)}
    # conv kernels of one bench step: per-dispatch average x dispatches per step (steps = dispatches of the VQ kernel)
    # steps of the bench run = dispatches of a kernel that runs exactly once per step (the decoder's last kernel; before
    # round 3's fusion the quantizer kernel served: it now runs only in bench.py's extra launches)
    once = [n for n in bf if "dec_tail8" in n or "convt_out_kernel" in n]
    bvq = [n for n in bf if is_vq(n)]
    steps = max(bf[n][1] for n in once) if once else (max(bf[n][1] for n in bvq) if bvq else 1)
    conv, table = 0.0, {}
    for n in sorted(bf):
        rd, cnt = bf[n]
        wr = bw.get(n, (0.0, 0))[0]
        per_step = cnt / steps
        byts = (2 * rd + wr) * 1024 * per_step
        table[n[:70]] = {"launches_per_step": round(per_step, 2), "read_KiB_x2": round(2 * rd), "write_KiB": round(wr)}
        # (kernels that run less than once per step belong to bench.py's extra launches -- warm-up packs, the unfused encoder in
        # front of the standalone quantizer measurement -- not to the step)
        if per_step >= 0.99 and any(t in n for t in ("conv_tile8", "res_tile8", "res_pair8", "conv_res_pair8", "res_layer", "conv_igemm",
                                                     "enc_front8", "dec_tail8", "conv_halo8", "res_halo8")):
            conv += byts
    # the quantizer per TEMPLATE INSTANCE and ROW COUNT (round 5): bench.py takes `roofline.traffic` from here for
    # the very instance it names and times -- the bench run's own extra launches of the stand-alone kernel at the workload's rows,
    # plus the beyond-the-Infinity-Cache stream of tools/vq_traffic.py
    import re
    import bench as _bench
    hw = _bench.WORKLOADS[wl][2]
    rows_wl = B * (hw // 4) * (hw // 4)
    inst = {}
    for tab_f, tab_w, rows in ((bf, bw, rows_wl), (vf, vw, vq_rows)):
        for n in tab_f:
            mm = re.search(r"(vq_\w+<[^>]*>)", n)
            if not mm or not is_vq(n):
                continue
            inst[f"{mm.group(1)}@{rows}"] = {"read_bytes": round(2 * tab_f[n][0] * 1024), "write_bytes": round(tab_w.get(n, (0.0, 0))[0] * 1024),
                                             "algorithmic_bytes": rows * (8 * _bench.WORKLOADS[wl][4] + 8), "dispatches": tab_f[n][1]}
    # the quantizer FAMILY of one bench step (the streamed-codebook kernels run four launches per slab of 2^18 rows: bench.py times them
    # together and names the family by its sweep kernel): every dispatch of a quantizer kernel in the bench passes, per step
    once_v = [n for n in bf if "dec_tail8" in n or "convt_out_kernel" in n]
    steps_v = max(bf[n][1] for n in once_v) if once_v else 1
    fam_r = sum(2 * bf[n][0] * 1024 * bf[n][1] / steps_v for n in bf if is_vq(n))
    fam_w = sum(bw.get(n, (0.0, 0))[0] * 1024 * bf[n][1] / steps_v for n in bf if is_vq(n))
    if fam_r + fam_w > 0:
        inst[f"vq_step@{rows_wl}"] = {"read_bytes": round(fam_r), "write_bytes": round(fam_w), "algorithmic_bytes": rows_wl * (8 * _bench.WORKLOADS[wl][4] + 8),
                                      "dispatches": sum(bf[n][1] for n in bf if is_vq(n)), "note": "all quantizer launches of one step"}
    for v in inst.values():
        v["over_algorithmic"] = round((v["read_bytes"] + v["write_bytes"]) / v["algorithmic_bytes"], 4)
    res["vq_instances"] = inst
    res["conv_bytes_per_image"] = round(conv / B, 1)     # on the fused 32x32 path this includes the quantizer's z_q / index writes
    # stamp: bench.py uses this file only while the kernel sources are the ones it was measured on
    import bench
    res["source_sha"] = bench.source_sha()
    try:
        res["git_head"] = subprocess.check_output(["git", "-C", bench.ROOT, "rev-parse", "--short=12", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        res["git_head"] = None
    rsrc = resources(sys.argv[3])
    res["kernel_resources"] = {k: v for k, v in rsrc.items() if "vqvae::" in k and "pack" not in k and "prepare" not in k}
    res["per_kernel_KiB_per_dispatch"] = table
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_kernel_KiB_per_dispatch"}))


if __name__ == "__main__":
    main()
