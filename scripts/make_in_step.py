"""Builds profiles/<tag>_in_step_kernels.json (+ the per-step breakdown text files and kernel-stats copies) from the output of
scripts/profile_round.sh:  python scripts/make_in_step.py gpurun_out r03
Per dtype: in-step launches / average duration of every hand-written kernel group bench.py's roofline probe reports, the
group on top of the step (by in-step time), and HBM traffic per launch of the probe's launches from the two PMC passes
(FETCH_SIZE x2 on gfx950 -- it tallies 128-byte requests at 64 bytes -- x1024; WRITE_SIZE x1024; MI355X_MICROARCH.md HBM)."""
import collections, csv, glob, json, os, re, shutil, sys

base, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = {
    "f32": {"acr_gemm_f32_tn": ["gemm_f32_dma_kernel<false, false, 3>", "gemm_f32_kernel<false, false, 3>", "gemm_f32_reduce"],
            # forward Linears AND input gradients (NT on the cached W^T since round 2): one symbol family
            "acr_gemm_f32_nt": ["gemm_f32_dma_kernel<true, true,", "gemm_f32_kernel<true, true,"],
            "acr_gemm_f32_nn": ["gemm_f32_dma_kernel<true, false,", "gemm_f32_kernel<true, false,"],
            "acr_attn_bwd": ["attn_delta_sres", "attn_bwd_sres_kernel", "attn_delta_dma_kernel", "attn_bwd_dma_kernel", "attn_dq_dma_kernel", "attn_dkdv_dma_kernel", "attn_delta_kernel<float>",
                             "attn_dq_kernel<float>", "attn_dkdv_kernel<float>"],
            "acr_attn_fwd": ["attn_fwd_sres_kernel", "attn_pmean_sres_kernel", "attn_fwd_dma_kernel", "attn_pmean_dma_kernel", "attn_fwd_kernel<float>", "attn_tile_qk_kernel<float, 0>"],
            "acr_consistency_fwd": ["cons_fwd", "cons_reduce"]},
    # split products (math = "f32_split"): products on images (gemm_f32_planes_*), the image pass, split-product attention
    "f32_split": {"acr_gemm_f32_tn": ["gemm_f32_planes_tn_kernel", "gemm_f32_reduce"],
                  "acr_gemm_f32_nt": ["gemm_f32_planes_kernel<"],
                  "acr_x3_image": ["planes_tile_kernel", "planes_tile_t_kernel", "planes_tile_many", "planes_colsum_kernel"],
                  "acr_attn_bwd": ["attn_delta_sres", "attn_bwd_x3_kernel"],
                  "acr_attn_fwd": ["attn_fwd_x3_kernel", "attn_pmean_sres_kernel"],
                  "acr_conv_stem": ["gemm_f32_split_kernel", "gemm_f32_wimg", "conv3x3_", "conv1x1_ksum", "c3_s2d", "c3_d2s", "c3_zero", "subsample2", "acr_slab_sum_wide"],
                  "acr_groupnorm": ["gnf_"],
                  "acr_consistency_fwd": ["cons_fwd", "cons_reduce"]},
    "bf16": {"acr_wgrad_bf16": ["gemm_tn_bf16", "wgrad_reduce"],
             "acr_linear_bf16": ["gemm_nt_bf16_wide_kernel<true", "gemm_nt_bf16_dma_kernel<true"],
             "acr_linear_bf16_dx": ["gemm_nt_bf16_wide_kernel<false", "gemm_nt_bf16_dma_kernel<false"],
             "acr_attn_bwd": ["attn_delta_bf16", "attn_dq_bf16", "attn_dq4_bf16", "attn_dkdv_bf16"],
             "acr_attn_fwd": ["attn_fwd_bf16", "attn_tile_qk_bf16_kernel<0>"],
             "acr_consistency_fwd": ["cons_fwd", "cons_reduce"]},
}
# what ONE launch of the group means in the probe (number of kernel symbols that make up the group's launch)
out = {}
for dt, groups in GROUPS.items():
    d = os.path.join(base, "%s_%s" % (tag, dt))
    tr = glob.glob(d + "/**/p_kernel_trace.csv", recursive=True)
    if not tr:
        continue
    rows = list(csv.DictReader(open(tr[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["Start_Timestamp"]) for r in rows if "cons_fwd" in r["Kernel_Name"]]
    real = [i for i in range(len(marks) - 1) if marks[i + 1] - marks[i] > 10e6]
    nsteps = 4
    t0, t1 = marks[real[-1 - nsteps]], marks[real[-1]]
    per = collections.defaultdict(lambda: [0, 0.0])
    busy = 0.0
    lines = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        s = int(r["Start_Timestamp"])
        if t0 <= s < t1:
            dur = (int(r["End_Timestamp"]) - s) / 1e3
            busy += dur
            n = r["Kernel_Name"]
            lines[re.sub(r"\(.*", "", n)[:90]][0] += 1
            lines[re.sub(r"\(.*", "", n)[:90]][1] += dur
            for key, pats in groups.items():
                if any(p in n for p in pats):
                    per[key][0] += 1
                    per[key][1] += dur
                    break
    rec = {"wall_ms_per_step": round((t1 - t0) / 1e6 / nsteps, 3), "kernel_busy_ms_per_step": round(busy / 1e3 / nsteps, 3), "kernels": {}}
    for key, (n, us) in per.items():
        rec["kernels"][key] = {"kernel_launches_per_step": n // nsteps, "ms_per_step": round(us / 1e3 / nsteps, 3), "symbols": groups[key]}
    rec["top"] = max(rec["kernels"], key=lambda k: rec["kernels"][k]["ms_per_step"])
    with open(os.path.join(ROOT, "profiles", "%s_step_breakdown_%s.txt" % (tag, dt)), "w") as f:
        f.write("rocprofv3 --kernel-trace -- python3 bench.py --dtype %s --steps 6 --warmup 3; last %d steady steps\n" % (dt, nsteps))
        f.write("wall %.2f ms/step, kernel busy %.2f ms/step\n" % (rec["wall_ms_per_step"], rec["kernel_busy_ms_per_step"]))
        for n, v in sorted(lines.items(), key=lambda kv: -kv[1][1])[:60]:
            f.write("%-90s %5d %8.3f ms/step\n" % (n, v[0] // nsteps, v[1] / 1e3 / nsteps))
    st = glob.glob(d + "/**/p_kernel_stats.csv", recursive=True)
    if st:
        shutil.copy(st[0], os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, dt)))
    # PMC traffic of the probe's launches
    traffic = {}
    for cn, mult in (("FETCH_SIZE", 2 * 1024), ("WRITE_SIZE", 1024)):
        f = glob.glob(os.path.join(base, "%s_pmc%s_%s" % (tag, cn[0], dt)) + "/**/p_counter_collection.csv", recursive=True)
        if not f:
            continue
        shutil.copy(f[0], os.path.join(ROOT, "profiles", "%s_pmc_%s_probe_%s.csv" % (tag, cn, dt)))
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == cn:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for key, pats in groups.items():
            tot = 0.0
            for kn, vals in acc.items():
                if any(p in kn for p in pats):
                    tot += sum(vals) / len(vals)           # one launch of each symbol of the group
            traffic.setdefault(key, {})[cn.lower() + "_bytes"] = int(tot * mult)
    for key, v in traffic.items():
        v["bytes_per_launch"] = v.get("fetch_size_bytes", 0) + v.get("write_size_bytes", 0)
    rec["traffic"] = traffic
    rec["traffic_note"] = "per launch of the roofline probe's geometry (fc1-shaped GEMMs; attention at B=32 views, H=12, T=785); FETCH_SIZE KB x2 x1024, WRITE_SIZE KB x1024 (MI355X_MICROARCH.md, HBM)"
    out[dt] = rec
dst = os.path.join(ROOT, "profiles", "%s_in_step_kernels.json" % tag)
if os.path.exists(dst):                                    # a run over a subset of the dtypes replaces those records only
    prev = json.load(open(dst))
    prev.update(out)
    out = prev
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: {"top": v["top"], "wall": v["wall_ms_per_step"]} for k, v in out.items()}))
