#!/usr/bin/env python3
"""Condense gpurun_out/prof/<tag>/ (tools/profile.sh output) into tracked files under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_pmc.json           per-kernel PMC averages + derived numbers
  profiles/traffic.json             HBM bytes per launch of the dominant kernel (read by bench.py)

HBM bytes follow MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are collected in
separate passes and are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a
coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
The doubling is checked here against a launch whose read volume is known (the aggregate kernel
reads exactly C * n * 16 bytes)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources  # noqa: E402


def newest(pattern):
    """gpurun merges every call's output into the same directory: take the most recent run's file."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1] if files else None


def rows(pattern):
    f = newest(pattern)
    return list(csv.DictReader(open(f))) if f else []


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", "prof", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        for r in rows(os.path.join(src, sub, "*", "*_counter_collection.csv")):
            k = r["Kernel_Name"]
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            # (rocprofv3's VGPR_Count / SGPR_Count columns are NOT the kernel's registers -- they read 52 / 112 for nearly every kernel
            # of this library; the resource columns below come from the code objects inside the library that ran: tools/kernel_resources.py)
            per[k]["_meta"] = [{"lds_bytes_dispatch": int(r["LDS_Block_Size"]), "workgroup": int(r["Workgroup_Size"]),
                                "grid": int(r["Grid_Size"])}]
    for r in sorted(rows(os.path.join(src, "trace", "*", "*_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"])):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))      # in launch order
    # which library the profiled run loaded: its bench line says so (config.library); the product library otherwise
    lib_name, steps_timed = "libflashe_hip.so", None
    try:
        for line in open(os.path.join(src, "trace.log")):
            if line.startswith('{"metric'):
                j = json.loads(line)
                lib_name, steps_timed = j.get("config", {}).get("library", lib_name), j.get("steps")
    except OSError:
        pass
    try:
        res = {kernel_resources.norm(k): v for k, v in kernel_resources.resources(os.path.join(ROOT, "flashe_amd", lib_name)).items()}
    except Exception as exc:                               # (no toolchain at hand: the columns are left out, never guessed)
        print(f"warning: no code-object resources ({exc})", file=sys.stderr)
        res = {}
    out = {}
    for k, d in per.items():
        if not k.startswith("void flashe::") and not k.startswith("flashe::"):
            continue
        e = {"launch": dict(d["_meta"][0])}
        co = res.get(kernel_resources.norm(k))
        if co:
            e["launch"].update({"vgpr": co["vgpr"], "agpr": co["agpr"], "sgpr": co["sgpr"], "scratch_bytes_per_lane": co["scratch_bytes_per_lane"],
                                "lds_bytes_static": co["lds_bytes_static"], "waves_per_simd_by_vgpr": kernel_resources.waves_per_simd(co["vgpr"], co["agpr"]),
                                "resource_source": f"code object inside flashe_amd/{lib_name} (llvm-readelf --notes)"})
        for c, v in d.items():
            if c == "_meta":
                continue
            e[c] = {"avg": sum(v) / len(v), "min": min(v), "max": max(v), "launches": len(v)}
        if k in dur:
            e["avg_duration_us_unprofiled_pass"] = sum(dur[k]) / len(dur[k]) / 1e3          # every launch of the process, cold ones included
            if steps_timed and len(dur[k]) >= steps_timed:
                # the bench's timed region is the END of the trace: the last K launches of a kernel that runs once per round
                e["avg_us_timed_launches"] = sum(dur[k][-steps_timed:]) / steps_timed / 1e3
                e["timed_launches"] = steps_timed
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch_avg"] = (2 * e["FETCH_SIZE"]["avg"] + e["WRITE_SIZE"]["avg"]) * 1024
        if "GRBM_GUI_ACTIVE" in e and k in dur:
            e["effective_clock_ghz"] = e["GRBM_GUI_ACTIVE"]["avg"] / 8 / (sum(dur[k]) / len(dur[k]))
        out[k] = e
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
    # HBM traffic per launch of every kernel of the round, and -- for the dominant one -- its ratio to the algorithmic bytes
    # the profiled bench run itself reported (bench.py multiplies that ratio by its own algorithmic bytes: roofline.traffic)
    alg, dom_name = None, None
    for log in ("pmc_fetch.log", "trace.log"):
        try:
            for line in open(os.path.join(src, log)):
                if line.startswith('{"metric'):
                    rl = json.loads(line)["roofline"]
                    alg, dom_name = rl["algorithmic_bytes_per_launch"], rl.get("kernel_key") or rl["kernel"].split("<")[0].split(" ")[0]
        except OSError:
            pass
        if alg:
            break
    traffic = {"tag": tag, "source": f"profiles/{tag}_pmc.json",
               "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of the default bench command; "
                      "bytes = (2 * FETCH_SIZE[KiB] + WRITE_SIZE[KiB]) * 1024 (gfx950 FETCH_SIZE counts 64 B per 128-B "
                      "request, MI355X_MICROARCH.md HBM section), averaged over the kernel's launches",
               "kernels": {}}
    for k, e in out.items():
        if "FETCH_SIZE" not in e or "WRITE_SIZE" not in e:
            continue
        short = k.replace("void ", "").replace("flashe::", "").split("<")[0].split("(")[0]
        if short == "prf_chain_kernel" and "prf_chain_kernel<1024, true" in k:
            short = "prf_chain_kernel_sum"          # the instantiation that also writes the local partial aggregate
        if short == "span_prf_kernel" and "span_prf_kernel<0>" in k:
            short = "span_prf_kernel_decrypt"       # (the ENC instantiations -- encrypts + aggregate -- keep the plain key)
        ent = {"full_name": k, "hbm_bytes_per_launch": e["hbm_bytes_per_launch_avg"], "fetch_kib_avg": e["FETCH_SIZE"]["avg"],
               "write_kib_avg": e["WRITE_SIZE"]["avg"], "launches": e["FETCH_SIZE"]["launches"]}
        if dom_name and short == dom_name and alg:
            ent["algorithmic_bytes_per_launch"] = alg
            ent["hbm_bytes_per_algorithmic_byte"] = e["hbm_bytes_per_launch_avg"] / alg
            ent["measured_in"] = f"profiles/{tag}_pmc.json"          # per-kernel provenance: a later tag rewrites the file's top level
        if short == "aggregate_elem_kernel":
            ent["note"] = "calibration: this reduce reads exactly C x what it writes, so 2 * FETCH / WRITE must equal C"
        prev = traffic["kernels"].get(short)
        if prev is None or ent["launches"] > prev["launches"]:
            traffic["kernels"][short] = ent
    # a ratio measured by an earlier round's passes stays available for the kernels this run did not time as its dominant one
    # (bench.py --no-partial-agg times the plain chained launch): carried over with its origin
    try:
        prev_t = json.load(open(os.path.join(dst, "traffic.json")))
        for key, ent in prev_t.get("kernels", {}).items():
            if "hbm_bytes_per_algorithmic_byte" in ent and "hbm_bytes_per_algorithmic_byte" not in traffic["kernels"].get(key, {}):
                keep = dict(ent)
                keep.setdefault("measured_in", prev_t.get("source"))
                if key in traffic["kernels"]:
                    keep["this_run_other_shape"] = traffic["kernels"][key]
                traffic["kernels"][key] = keep
    except (OSError, ValueError):
        pass
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    # The --stats average of the dominant kernel covers every launch of the process -- parity run, settle rounds, warmup (the first
    # ones after the host-side parity check run on a cold clock) and the K timed ones.  bench.py's roofline.avg_launch_ms covers the K
    # timed launches only, which are the LAST K of the trace: put both averages side by side.
    if dom_name:
        steps, bench_ms = None, None
        try:
            for line in open(os.path.join(src, "trace.log")):
                if line.startswith('{"metric'):
                    j = json.loads(line)
                    steps, bench_ms = j["steps"], j["roofline"]["avg_launch_ms"]
        except OSError:
            pass
        def is_dom(k):
            if dom_name == "prf_chain_kernel_sum":
                return "prf_chain_kernel<1024, true" in k
            if dom_name == "prf_chain_kernel":
                return "prf_chain_kernel<1024, false" in k
            if dom_name == "span_prf_kernel":
                return "span_prf_kernel<1>" in k or "span_prf_kernel<2>" in k
            return dom_name in k
        d_all = [v for k, vs in dur.items() if is_dom(k) for v in vs]
        if steps and len(d_all) >= steps:
            json.dump({"kernel": dom_name, "launches_in_trace": len(d_all), "avg_us_all_launches": sum(d_all) / len(d_all) / 1e3,
                       "timed_launches": steps, "avg_us_last_timed_launches": sum(d_all[-steps:]) / steps / 1e3,
                       "bench_roofline_avg_launch_us_same_run": bench_ms * 1e3,
                       "first_five_us": [v / 1e3 for v in d_all[:5]]},
                      open(os.path.join(dst, f"{tag}_timed_launches.json"), "w"), indent=1)
    print(json.dumps({k[:50]: {c: (round(v["avg"]) if isinstance(v, dict) and "avg" in v else v) for c, v in e.items()
                               if c != "launch"} for k, e in out.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r04")
