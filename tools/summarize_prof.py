"""Summarise gpurun_out/prof (tools/gpu_profile.sh, tools/gpu_r4_profile.sh) into profiles/: kernel stats, PMC counters, HBM traffic json."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
KERNEL = "smj_step_kernel"


def render_summary(out):
    """Ray-casting kernels (tools/gpu_render_prof.py): stats + per-kernel PMC means."""
    path = os.path.join(SRC, "rtrace", "smj_kernel_stats.csv")
    if not os.path.exists(path):
        return
    out.append("\n## Depth / lidar kernels: `rocprofv3 --kernel-trace --stats -- python tools/gpu_render_prof.py` 4096 stretch_kitchen_robocasa` (the kitchen at Robocasa scale, 4096 envs, both depth cameras + lidar; rounds 3-5: the kitchen stand-in)\n")
    out.append("A depth render of one camera = `smj_depth_prepass` (per-env staging) + `smj_fill_kernel` (z-buffer) + `smj_meshlet_kernel` (meshes and boxes rasterised, "
               "atomicMin) + `smj_depth_kernel<false, true>` (per-pixel: remaining primitives, handed-over triangles, limits); calls 1 and 3 of each are the one-off camera-static layers.\n")
    out.append("| kernel | calls | total ms | avg ms | % | min ms | max ms |\n|---|---|---|---|---|---|---|")
    with open(path) as f:
        for r in csv.DictReader(f):
            if any(k in r["Name"] for k in ("smj_depth", "smj_meshlet", "smj_fill", "smj_lidar", "smj_stage", "smj_base")):
                out.append(f"| `{r['Name'][:50]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e6:.3f} | {float(r['Percentage']):.2f} | {float(r['MinNs'])/1e6:.3f} | {float(r['MaxNs'])/1e6:.3f} |")
    per = collections.defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(SRC, "rpmc_*", "smj_counter_collection.csv"))):
        acc = collections.defaultdict(list)
        with open(d) as f:
            for r in csv.DictReader(f):
                for kn in ("smj_meshlet_kernel", "smj_depth_kernel", "smj_depth_prepass", "smj_lidar_kernel"):
                    if kn in r["Kernel_Name"]:
                        acc[(kn, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (kn, cn), v in acc.items():
            v = [x for x in v if x > 0.05 * max(v)] if kn in ("smj_depth_kernel", "smj_meshlet_kernel") and len(v) > 4 else v   # skip the two one-off camera-static layer renders
            per[kn][cn] = sum(v) / len(v)
    out.append("\nPMC means per dispatch (separate `--pmc` passes):\n")
    names = sorted({c for k in per.values() for c in k})
    out.append("| counter | " + " | ".join(per.keys()) + " |\n|---|" + "---|" * len(per))
    for c in names:
        out.append(f"| {c} | " + " | ".join(f"{per[k].get(c, float('nan')):.4g}" for k in per) + " |")
    for kn, v in per.items():
        bits = []
        if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU"):
            bits.append(f"VALU lane utilisation {v['SQ_THREAD_CYCLES_VALU'] / (64 * v['SQ_ACTIVE_INST_VALU']):.2f}")
        if v.get("SQ_BUSY_CYCLES") and v.get("SQ_ACTIVE_INST_VALU"):
            bits.append(f"VALU-active / wave-cycles {v['SQ_ACTIVE_INST_VALU'] / max(1.0, 4 * v.get('SQ_WAVE_CYCLES', 1)):.2f}")
        if v.get("TCC_HIT_sum") is not None and (v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) > 0:
            bits.append(f"L2 hit rate {v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.3f}")
        if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
            bits.append(f"HBM: FETCH_SIZE {v.get('FETCH_SIZE', 0) * 1024 / 1e6:.1f} MB (x2 for wide streams), WRITE_SIZE {v.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB per dispatch")
        out.append(f"\n`{kn}`: " + "; ".join(bits))


def satellite_summary(out):
    """Satellite builds (tools/gpu_r4_profile.sh): kernel stats of the Robocasa-scale kitchen, kitchen4 and scene.xml on smj_step_kernel_sat2 (Newton) / _satp (PGS),
    PMC means of the kitchen's launches."""
    for sub, log, title in (("rctrace", "rc_trace.log", "kitchen at Robocasa scale (`tools/gpu_options_probe.py scene=stretch_kitchen_robocasa`: 44 fixture bodies, 307 collision geoms, 8 articulated fixture parts + 8 free objects = 16 satellites; 4096 envs; primary kernel `smj_step_kernel_sat2` (two wavefronts per env, round 5), two envs per CU; parked chunks go to `smj_step_kernel_sat32_worker`: pollers beside the launch + the sweep after it)"),
                            ("rcpgstrace", "rcpgs_trace.log", "the same kitchen under PGS (`tools/gpu_options_probe.py scene=stretch_kitchen_robocasa solver=0`; `smj_step_kernel_satp` = the 16-satellite build's PGS-only kernel, TWO wavefronts per env: the satellite islands swept beside the dense system; second start from the previous step's forces on)"),
                            ("trace_stretch_kitchen4_sat", "trace_stretch_kitchen4_sat.log", "kitchen with four free objects on the satellite build (`scene=stretch_kitchen4_sat`)"),
                            ("trace_stretch_scene_sat", "trace_stretch_scene_sat.log", "the reference's scene.xml on the satellite build (`scene=stretch_scene_sat`)")):
        path = os.path.join(SRC, sub, "smj_kernel_stats.csv")
        if not os.path.exists(path):
            continue
        out.append(f"\n## `rocprofv3 --kernel-trace --stats`: {title}\n")
        out.append("| kernel | calls | total ms | avg ms | % | min ms | max ms |\n|---|---|---|---|---|---|---|")
        with open(path) as f:
            for r in csv.DictReader(f):
                if float(r["Percentage"]) > 0.05:
                    out.append(f"| `{r['Name'][:60]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e6:.3f} | {float(r['Percentage']):.2f} | {float(r['MinNs'])/1e6:.3f} | {float(r['MaxNs'])/1e6:.3f} |")
        lp = os.path.join(SRC, log)
        if os.path.exists(lp):
            last = [l for l in open(lp) if "env-steps/s" in l]
            if last:
                out.append("\nProbe output under the profiler: `" + last[-1].strip()[:200] + "`")
        tp = os.path.join(SRC, sub, "smj_kernel_trace.csv")
        if os.path.exists(tp):
            with open(tp) as f:
                allk = list(csv.DictReader(f))
            kn = "smj_step_kernel_satp(" if sub == "rcpgstrace" else "smj_step_kernel_sat2("
            tr = [r for r in allk if r["Kernel_Name"].startswith(kn)]
            if tr:
                dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in tr]
                r0 = tr[0]
                out.append(f"\nPer-dispatch durations of `{kn[:-1]}` (ms; the first is the 300-step settle, then launches of 50 steps: "
                           f"6 settled + 14 under random actions): {[round(x, 1) for x in dur]}")
                out.append(f"Resources: VGPR {r0['VGPR_Count']} (+AGPR {r0['Accum_VGPR_Count']}), SGPR {r0['SGPR_Count']}, scratch {r0['Scratch_Size']} B, "
                           f"workgroup {r0['Workgroup_Size_X']}, grid {r0['Grid_Size_X']} (dynamic LDS per env: 81 904 B with one wavefront, 81 920 B -- the mailbox -- with two: two envs per CU).")
    scenes_traffic = {}
    for pre, kn, what in (("rc", "smj_step_kernel_sat2(", "Newton, two wavefronts per env"), ("rcpgs", "smj_step_kernel_satp(", "PGS, two wavefronts per env")):
        kv = {}
        for d in sorted(glob.glob(os.path.join(SRC, pre + "pmc_*", "smj_counter_collection.csv"))):
            acc = collections.defaultdict(list)
            with open(d) as f:
                for r in csv.DictReader(f):
                    if r["Kernel_Name"].startswith(kn):
                        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for name, v in acc.items():
                v = v[-10:]
                kv[name] = sum(v) / len(v)
        if kv.get("FETCH_SIZE") or kv.get("WRITE_SIZE"):
            scenes_traffic["stretch_kitchen_robocasa" + (":pgs" if pre == "rcpgs" else "")] = {
                "kernel": kn[:-1], "envs_per_gpu": 4096, "steps_per_launch": 50, "hbm_bytes_per_launch": 2 * kv.get("FETCH_SIZE", 0) * 1024 + kv.get("WRITE_SIZE", 0) * 1024,
                "fetch_bytes_raw": kv.get("FETCH_SIZE", 0) * 1024, "write_bytes_raw": kv.get("WRITE_SIZE", 0) * 1024}
        if kv:
            out.append(f"\n## PMC counters of `{kn[:-1]}` (Robocasa-scale kitchen, 4096 envs, {what}, 50-step launches under random actions; each group its own `--pmc` run), per launch, last 10 launches\n")
            out.append("| counter | mean |\n|---|---|")
            for name in sorted(kv):
                out.append(f"| {name} | {kv[name]:.4g} |")
            wc = kv.get("SQ_WAVE_CYCLES", 0)
            if wc:
                issued = kv.get("SQ_INSTS_VALU", 0) + kv.get("SQ_INSTS_SALU", 0) + kv.get("SQ_INSTS_LDS", 0)
                nq, nv = 29 + 8 + 8 * 7, 28 + 8 + 8 * 6   # robot + 8 single-joint parts + 8 free objects
                words = 2 * (nq + nv) + nv + 10 + nv   # qpos, qvel in and out; warm start; ctrl; act
                out.append(f"\n{4 * wc / max(1, issued):.1f} shader cycles per issued instruction (two wavefronts per env, two envs per CU: one wave on each of the CU's four SIMDs; an env's second wavefront sleeps at a workgroup barrier between its jobs, and its cycles count here); "
                           f"MFMA busy {kv.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1, 4 * wc):.4f}; LDS bank-conflict ratio {kv.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, kv.get('SQ_ACTIVE_INST_LDS', 1)):.3f}; "
                           f"VMEM reads per launch {kv.get('SQ_INSTS_VMEM_RD', 0):.3g} (static-geometry records and pair tables come from L2 / HBM, not LDS); "
                           f"HBM per launch: FETCH_SIZE {kv.get('FETCH_SIZE', 0) * 1024 / 1e6:.1f} MB + WRITE_SIZE {kv.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB "
                           f"(state moved once per launch: about {words} words x 4 B x 4096 envs = {words * 4 * 4096 / 1e6:.1f} MB; the model blob is 8 MB and stays in L2 / MALL; the writes beyond the state are manifold-cache entries -- 160 B per narrowphase miss, DevState::mcache -- and scratch spill stores: at the launch's duration a few GB/s, two orders of magnitude under the HBM roof -- this kernel is latency-bound like the standard one).")
    if scenes_traffic:
        with open(os.path.join(DST, "pmc_traffic_scenes.json"), "w") as f:
            json.dump({"scenes": scenes_traffic, "source": f"profiles/{TAG}_rocprof_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of tools/gpu_options_probe.py scene=stretch_kitchen_robocasa [solver=0], "
                                                              "separate passes, last 10 launches under random actions; read side x2 per MI355X_MICROARCH.md section HBM)"}, f, indent=1)
    for scn in ("stretch_kitchen_robocasa", "stretch_kitchen4_sat", "stretch_scene_sat", "stretch_kitchen4", "stretch_scene"):
        scp = os.path.join(ROOT, "gpurun_out", f"stage_cycles_{scn}.txt")
        if os.path.exists(scp):
            with open(scp) as fi, open(os.path.join(DST, f"{TAG}_stage_cycles_{scn}.txt"), "w") as fo:
                fo.write(fi.read())
    path = os.path.join(SRC, "trace_pgs_kitchen4_sat", "smj_kernel_stats.csv")
    if os.path.exists(path):
        out.append("\n## `rocprofv3 --kernel-trace --stats`: PGS with constraint islands, `stretch_kitchen4_sat` (`tools/gpu_options_probe.py solver=0 scene=stretch_kitchen4_sat`; `smj_step_kernel_satp` = the 16-satellite build with TWO wavefronts per env: grid x 128 threads)\n")
        out.append("| kernel | calls | total ms | avg ms | % | min ms | max ms |\n|---|---|---|---|---|---|---|")
        with open(path) as f:
            for r in csv.DictReader(f):
                if float(r["Percentage"]) > 0.05:
                    out.append(f"| `{r['Name'][:60]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e6:.3f} | {float(r['Percentage']):.2f} | {float(r['MinNs'])/1e6:.3f} | {float(r['MaxNs'])/1e6:.3f} |")
        lp = os.path.join(SRC, "trace_pgs_kitchen4_sat.log")
        if os.path.exists(lp):
            last = [l for l in open(lp) if "env-steps/s" in l]
            if last:
                out.append("\nProbe output under the profiler: `" + last[-1].strip()[:200] + "`")
        tp = os.path.join(SRC, "trace_pgs_kitchen4_sat", "smj_kernel_trace.csv")
        with open(tp) as f:
            tr = [r for r in csv.DictReader(f) if r["Kernel_Name"].startswith("smj_step_kernel_satp(")]
        if tr:
            r0 = tr[0]
            out.append(f"Resources of `smj_step_kernel_satp`: VGPR {r0['VGPR_Count']} (+AGPR {r0['Accum_VGPR_Count']}), scratch {r0['Scratch_Size']} B, workgroup {r0['Workgroup_Size_X']}, grid {r0['Grid_Size_X']}.")
    for name in ("pgs_stage_cycles_stretch_kitchen4_sat.txt", "pgs_stage_cycles_stretch_kitchen_robocasa.txt", "pgs_sat_probe.txt", "step1_probe.txt", "step1_trace.txt"):
        src = os.path.join(ROOT, "gpurun_out", name)
        if os.path.exists(src):
            with open(src) as fi, open(os.path.join(DST, f"{TAG}_{name}"), "w") as fo:
                fo.write(fi.read())
    src = os.path.join(ROOT, "gpurun_out", "steplen.txt")
    if os.path.exists(src):
        with open(src) as fi, open(os.path.join(DST, f"{TAG}_step_length.txt"), "w") as fo:
            fo.write(fi.read())
    for name in ("sat_caps.txt", "scene_probes.txt", "soak.txt"):
        src = os.path.join(ROOT, "gpurun_out", name)
        if os.path.exists(src):
            with open(src) as fi, open(os.path.join(DST, f"{TAG}_{name}"), "w") as fo:
                fo.write(fi.read())


def main():
    os.makedirs(DST, exist_ok=True)
    out = [f"# {TAG}: rocprofv3 evidence for `python bench.py --no-second-solver --no-cpu-baseline --no-extra` (4096 envs, Newton, 50 steps/launch)\n"]
    with open(os.path.join(SRC, "trace", "smj_kernel_stats.csv")) as f:
        rows = list(csv.DictReader(f))
    k = [r for r in rows if r["Name"].startswith(KERNEL + "(")][0]
    out.append("## `rocprofv3 --kernel-trace --stats` (all kernels with > 0.01 % of GPU time)\n")
    out.append("| kernel | calls | total ms | avg ms | % | min ms | max ms |\n|---|---|---|---|---|---|---|")
    for r in rows:
        if float(r["Percentage"]) > 0.01:
            name = r["Name"][:60]
            out.append(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e6:.3f} | {float(r['Percentage']):.2f} | {float(r['MinNs'])/1e6:.3f} | {float(r['MaxNs'])/1e6:.3f} |")
    out.append("\n(`smj_step_kernel_tall_worker` is not a second hot kernel: its time is that of the two POLLER workgroups that stay resident beside "
               "each standard launch while a recent launch had a capacity escalation -- they idle-poll the escalation list -- plus the sweep launch that follows; "
               "their share of the GPU is 2 of ~1000 wave slots.)")
    with open(os.path.join(SRC, "trace", "smj_kernel_trace.csv")) as f:
        allk = list(csv.DictReader(f))
    is_std = lambda name: name.startswith(KERNEL + "(")
    tr = [r for r in allk if is_std(r["Kernel_Name"])]
    tall = [r for r in allk if r["Kernel_Name"].startswith(KERNEL + "_tall(")]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in tr]
    out.append(f"\nPer-dispatch durations of `{KERNEL}` (ms): {[round(x, 1) for x in dur]}")
    out.append("(dispatches 1-10: settle at the home keyframe; 11-14 untimed random-action pre-roll; 15-16 warm-up; the last 10 the timed region; "
               "grid = 4096 envs x 10 chunks of 5 steps -- pipelined chunks, DESIGN.md section 3)")
    # the tall variant's dispatches: a POLLER runs beside a standard launch (second stream: it starts before the standard kernel ends),
    # the SWEEP follows it on the main stream and finishes what is left of the escalation list
    span = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in tr]
    pollers, sweeps = [[] for _ in tr], [[] for _ in tr]
    for r in tall:
        t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        done = False
        for i, (a, b) in enumerate(span):
            if min(t1, b) - max(t0, a) > 0.5 * (t1 - t0):
                pollers[i].append((t1 - t0) / 1e6); done = True; break
        if not done:
            prev = [i for i, (a, b) in enumerate(span) if b <= t0]
            if prev:
                sweeps[prev[-1]].append((t1 - t0, t1))
    wall = []
    for i, (a, b) in enumerate(span):
        end = max([b] + [t1 for _, t1 in sweeps[i]])
        wall.append((end - a) / 1e6)
    if tall:
        out.append(f"\n`{KERNEL}_tall` beside / after each standard launch (capacity escalation): pollers resident for "
                   f"{[round(sum(x), 1) for x in pollers]} ms (0 = they left at once: no escalation in the last 8 launches), sweep "
                   f"{[round(sum(d for d, _ in x) / 1e6, 2) for x in sweeps]} ms.")
    timed, tw = dur[-10:], wall[-10:]
    out.append(f"Timed-region average: **{sum(timed)/len(timed):.2f} ms** per standard launch of 4096 envs x 50 steps; with the sweep that "
               f"follows it **{sum(tw)/len(tw):.2f} ms** from the start of the standard kernel to the end of the sweep (the pollers run "
               f"inside that span; bench.py's HIP events bracket it plus the staging transposes and the order kernel).")
    r0 = tr[0]
    out.append(f"\nResources: VGPR {r0['VGPR_Count']} (+AGPR {r0['Accum_VGPR_Count']}), SGPR {r0['SGPR_Count']}, LDS {r0['LDS_Block_Size']} B, "
               f"scratch {r0['Scratch_Size']} B, workgroup {r0['Workgroup_Size_X']}, grid {r0['Grid_Size_X']}.  (The trace lists static LDS only: "
               f"the step kernel's LDS is dynamic, 40 956 B per workgroup, Newton and PGS alike -- smj_lds_bytes(); the code "
               f"object's metadata gives 256 VGPR + 256 AGPR.)\n")
    out.append("## PMC counters (each group collected in its own `rocprofv3 --pmc ... --kernel-trace` run), per launch, timed region\n")
    out.append("| counter | mean | min | max |\n|---|---|---|---|")
    vals = {}
    for d in sorted(glob.glob(os.path.join(SRC, "pmc_*", "smj_counter_collection.csv"))):
        acc = collections.defaultdict(list)
        with open(d) as f:
            for r in csv.DictReader(f):
                if r["Kernel_Name"].startswith(KERNEL + "("):
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name, v in acc.items():
            v = v[-10:]
            vals[name] = sum(v) / len(v)
            out.append(f"| {name} | {vals[name]:.4g} | {min(v):.4g} | {max(v):.4g} |")
    fetch, write = vals.get("FETCH_SIZE", 0) * 1024, vals.get("WRITE_SIZE", 0) * 1024
    alg = 672 * 4096 * 50
    moved = (672 + 92 + 4 * 12 * 20) * 4096   # state in + out, joint readout, body poses of the last step: what one launch moves
    out.append(f"\nHBM traffic per launch (3 dispatches: import transposes, `smj_step_kernel`, export transposes; the step kernel's "
               f"share is listed): FETCH_SIZE {fetch/1e6:.2f} MB + WRITE_SIZE {write/1e6:.2f} MB = {(fetch+write)/1e6:.2f} MB "
               f"(KB counters x 1024, MI355X_MICROARCH.md section HBM; with the guide's x2 on the read side {(2*fetch+write)/1e6:.2f} MB).  "
               f"Algorithmic bytes per launch (672 B x 4096 envs x 50 steps) = {alg/1e6:.1f} MB; the state stays in LDS for the 50 "
               f"steps of a launch, so what a launch has to move is {moved/1e6:.2f} MB (state in and out once, readout, body poses).  "
               f"The step kernel reads and writes contiguous env-major rows (DevState::stage); round 1 measured 60.7 MB per launch with "
               f"lane = dim accesses on the batch-major arrays.")
    wc, busy = vals.get("SQ_WAVE_CYCLES", 0), vals.get("SQ_BUSY_CYCLES", 0)
    if wc:
        out.append(f"\nIssue mix per launch: VALU {vals.get('SQ_INSTS_VALU',0):.3g}, SALU {vals.get('SQ_INSTS_SALU',0):.3g}, LDS {vals.get('SQ_INSTS_LDS',0):.3g}, "
                   f"VMEM rd {vals.get('SQ_INSTS_VMEM_RD',0):.3g}, MFMA {vals.get('SQ_INSTS_MFMA',0):.3g} wave-instructions over {wc:.3g} wave-cycles "
                   f"(quad-cycles): {4*wc/max(1,vals.get('SQ_INSTS_VALU',1)+vals.get('SQ_INSTS_SALU',0)+vals.get('SQ_INSTS_LDS',0)):.1f} shader cycles per issued "
                   f"instruction at 1 wave/SIMD -- the kernel is bound by its own dependent-instruction latency, not by HBM, LDS bandwidth or the matrix cores "
                   f"(SQ_VALU_MFMA_BUSY_CYCLES / (4*SQ_WAVE_CYCLES) = {vals.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(1,4*wc):.4f}; "
                   f"SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = {vals.get('SQ_LDS_BANK_CONFLICT',0)/max(1,vals.get('SQ_ACTIVE_INST_LDS',1)):.3f}).")
    with open(os.path.join(DST, f"{TAG}_rocprof_summary.md"), "w") as f:
        f.write("\n".join(out) + "\n")
    with open(os.path.join(DST, "pmc_traffic.json"), "w") as f:
        json.dump({"kernel": KERNEL, "envs_per_gpu": 4096, "steps_per_launch": 50, "solver": "newton",
                   "hbm_bytes_per_launch": 2 * fetch + write, "fetch_bytes_raw": fetch, "write_bytes_raw": write,
                   "source": f"profiles/{TAG}_rocprof_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                             f"timed-region launches; read side x2 per MI355X_MICROARCH.md section HBM)"}, f, indent=1)
    render_summary(out)
    for sub, title in (("strace", "the reference's own scene (`tools/gpu_options_probe.py scene=stretch_scene`: table + 2 free objects, 38 dofs; `smj_step_kernel_big38`, two envs per CU, escalation target `smj_step_kernel_big_worker`)"),
                       ("k4trace", "kitchen with four free objects (`scene=stretch_kitchen4`, 50 dofs; `smj_step_kernel_big50`, two envs per CU)"),
                       ("ktrace", "kitchen stand-in (`tools/gpu_options_probe.py scene=stretch_kitchen_standin`: 300 settle steps, 14 random-action launches of 50 steps; primary kernel `smj_step_kernel_mid`: the 128-row build of the tall variant, three envs per CU)"),
                       ("ptrace", "PGS (`tools/gpu_options_probe.py solver=0`, empty scene, same schedule)")):
        path = os.path.join(SRC, sub, "smj_kernel_stats.csv")
        if not os.path.exists(path):
            continue
        out.append(f"\n## `rocprofv3 --kernel-trace --stats`: {title}\n")
        out.append("| kernel | calls | total ms | avg ms | % | min ms | max ms |\n|---|---|---|---|---|---|---|")
        with open(path) as f:
            for r in csv.DictReader(f):
                if float(r["Percentage"]) > 0.05:
                    out.append(f"| `{r['Name'][:60]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e6:.3f} | {float(r['Percentage']):.2f} | {float(r['MinNs'])/1e6:.3f} | {float(r['MaxNs'])/1e6:.3f} |")
        log = os.path.join(SRC, {"ktrace": "kitchen_trace.log", "ptrace": "pgs_trace.log", "strace": "scene_trace.log", "k4trace": "kitchen4_trace.log"}[sub])
        if os.path.exists(log):
            last = [l for l in open(log) if "env-steps/s" in l]
            if last:
                out.append("\nProbe output under the profiler: `" + last[-1].strip()[:200] + "`")
    # PMC passes over the kitchen4 workload (smj_step_kernel_big50)
    kv = {}
    for d in sorted(glob.glob(os.path.join(SRC, "k4pmc_*", "smj_counter_collection.csv"))):
        acc = collections.defaultdict(list)
        with open(d) as f:
            for r in csv.DictReader(f):
                if r["Kernel_Name"].startswith("smj_step_kernel_big50("):
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name, v in acc.items():
            v = v[-10:]
            kv[name] = sum(v) / len(v)
    if kv:
        out.append("\n## PMC counters of `smj_step_kernel_big50` (kitchen with four free objects, 4096 envs, 50-step launches; each group its own `--pmc` run), per launch, last 10 launches\n")
        out.append("| counter | mean |\n|---|---|")
        for name in sorted(kv):
            out.append(f"| {name} | {kv[name]:.4g} |")
        wc = kv.get("SQ_WAVE_CYCLES", 0)
        if wc:
            issued = kv.get("SQ_INSTS_VALU", 0) + kv.get("SQ_INSTS_SALU", 0) + kv.get("SQ_INSTS_LDS", 0)
            out.append(f"\n{4 * wc / max(1, issued):.1f} shader cycles per issued instruction (one wave per SIMD, two of a CU's four SIMDs occupied: 81.6 KB of LDS per env); "
                       f"MFMA busy {kv.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1, 4 * wc):.4f}; LDS bank-conflict ratio {kv.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, kv.get('SQ_ACTIVE_INST_LDS', 1)):.3f}; "
                       f"HBM per launch: FETCH_SIZE {kv.get('FETCH_SIZE', 0) * 1024 / 1e6:.1f} MB + WRITE_SIZE {kv.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB "
                       f"(algorithmic: (55 + 50 + 10 + 50 + 55 + 50 + 50) words x 4 B x 4096 envs x 50 steps = {(55 + 50 + 10 + 50 + 55 + 50 + 50) * 4 * 4096 * 50 / 1e6:.0f} MB; moved once per launch: 1/50 of that).")
    satellite_summary(out)
    with open(os.path.join(DST, f"{TAG}_rocprof_summary.md"), "w") as f:
        f.write("\n".join(out) + "\n")
    for scn in ("stretch_scene", "stretch_kitchen4"):
        scp = os.path.join(ROOT, "gpurun_out", f"stage_cycles_{scn}.txt")
        if os.path.exists(scp):
            with open(scp) as fi, open(os.path.join(DST, f"{TAG}_stage_cycles_{scn}.txt"), "w") as fo:
                fo.write(fi.read())
    sc = os.path.join(ROOT, "gpurun_out", "stage_cycles.txt")
    if os.path.exists(sc):   # tools/gpu_diag.py: per-stage shader cycles and event counts per env-step (profiling build of the standard kernel)
        with open(sc) as fi, open(os.path.join(DST, f"{TAG}_stage_cycles.txt"), "w") as fo:
            fo.write(fi.read())
    for name in ("smj_kernel_stats.csv", "smj_domain_stats.csv"):
        src = os.path.join(SRC, "trace", name)
        if os.path.exists(src):
            with open(src) as fi, open(os.path.join(DST, f"{TAG}_{name}"), "w") as fo:
                fo.write(fi.read())
    print("\n".join(out))


if __name__ == "__main__":
    main()
