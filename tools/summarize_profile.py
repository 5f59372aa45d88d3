#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace --stats, and separate --pmc FETCH_SIZE / WRITE_SIZE passes) into the
small text / JSON summaries committed under profiles/.

  python tools/summarize_profile.py gpurun_out/prof_r1 profiles/r1

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): bytes = counter * 1024, and
on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read stream, so the read side is doubled
("corrected"); both raw and corrected figures are written.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amber_amd._lib import csrc_hash  # noqa: E402


def short(name):
    for key in ("edge_kernel", "node_kernel", "embed_kernel", "pack_kernel", "csr_", "segment_mean", "permute_rows"):
        if key in name:
            i = name.find(key)
            j = name.find("(", i)
            return name[i:j if j > 0 else None]
    return name[:60]


def main(src, dst_prefix):
    os.makedirs(os.path.dirname(dst_prefix) or ".", exist_ok=True)
    lines = []
    stats = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)
    summary = {}
    if stats:
        lines.append("# rocprofv3 --kernel-trace --stats (bench.py --steps 3 --warmup 1 --no-cpu-baseline)")
        lines.append(f"{'kernel':<28}{'calls':>7}{'total_ms':>12}{'avg_ms':>10}{'min_ms':>10}{'max_ms':>10}{'pct':>8}")
        for row in csv.DictReader(open(stats[0])):
            name = short(row["Name"])
            lines.append(f"{name:<28}{int(row['Calls']):>7}{float(row['TotalDurationNs']) / 1e6:>12.3f}"
                         f"{float(row['AverageNs']) / 1e6:>10.4f}{float(row['MinNs']) / 1e6:>10.4f}{float(row['MaxNs']) / 1e6:>10.4f}"
                         f"{float(row['Percentage']):>8.2f}")
            summary[name] = {"calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6}
    pmc = {}
    for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        files = glob.glob(os.path.join(src, sub, "**", "*_counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = defaultdict(list)
        regs = {}
        for row in csv.DictReader(open(files[0])):
            if row["Counter_Name"] != counter:
                continue
            name = short(row["Kernel_Name"])
            acc[name].append(float(row["Counter_Value"]))
            regs[name] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Scratch_Size"])
        for name, vals in acc.items():
            pmc.setdefault(name, {})[counter] = sum(vals) / len(vals) * 1024.0  # bytes per launch
            pmc[name]["launches"] = len(vals)
            pmc[name]["regs"] = regs[name]
    if pmc:
        lines.append("")
        lines.append("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), average per launch")
        lines.append(f"{'kernel':<28}{'launches':>9}{'fetch_GB_raw':>14}{'fetch_GB_x2':>13}{'write_GB':>10}{'hbm_GB_corrected':>18}"
                     f"   vgpr/agpr/sgpr/lds/scratch")
        for name, v in sorted(pmc.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0))):
            f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
            lines.append(f"{name:<28}{v['launches']:>9}{f / 1e9:>14.3f}{2 * f / 1e9:>13.3f}{w / 1e9:>10.3f}{(2 * f + w) / 1e9:>18.3f}"
                         f"   {'/'.join(v['regs'])}")
            summary.setdefault(name, {}).update({"fetch_bytes_raw": f, "fetch_bytes_corrected": 2 * f, "write_bytes": w,
                                                 "hbm_bytes_corrected": 2 * f + w})
    text = "\n".join(lines) + "\n"
    open(dst_prefix + "_rocprof_summary.txt", "w").write(text)
    json.dump(summary, open(dst_prefix + "_rocprof_summary.json", "w"), indent=1)
    print(text)
    edge = [k for k in summary if k.startswith("edge_kernel")]
    if edge and "hbm_bytes_corrected" in summary[edge[0]]:
        json.dump({"edge_kernel_hbm_bytes_per_launch": summary[edge[0]]["hbm_bytes_corrected"],
                   "source": os.path.basename(dst_prefix) + "_rocprof_summary.txt", "csrc_hash": csrc_hash(),
                   "note": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024, averaged over the edge-kernel launches of one forward"},
                  open(os.path.join(os.path.dirname(dst_prefix) or ".", "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
