#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into profiles/<tag>_*:

    python tools/pmc_summary.py <tag> <trace_dir> <pmc_fetch_dir> <pmc_write_dir> [local_qubits of the profiled run, default 28]

  * <tag>_kernel_stats.csv   copy of rocprofv3 --kernel-trace --stats summary
  * <tag>_pmc_summary.json   per kernel class: launches, avg duration (from the stats pass),
                             HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
    (FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of a wide
    coalesced read stream -- MI355X_MICROARCH.md, HBM section -- hence the factor 2; the two
    counters are collected in separate --pmc passes).
"""
import collections
import csv
import glob
import json
import re
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from quantum_simulations_amd._lib import source_hash  # noqa: E402


def klass(name: str):
    m = re.search(r"k_gate_shuffle<(\d), (\d)", name)
    if m:
        return f"k_gate_shuffle<{m.group(1)},{m.group(2)}>"
    m = re.search(r"k_gate<(\d),", name)
    if m:
        return f"k_gate<{m.group(1)}>"
    m = re.search(r"k_dense_mfma2<(\d)", name)
    if m:
        return f"k_dense_mfma2<{m.group(1)}>"
    m = re.search(r"k_tile", name)
    if m:
        return "k_tile"
    return None


def counter_avg(directory: str, counter: str):
    out = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{directory}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = klass(r["Kernel_Name"])
            if k:
                out[k][0] += 1
                out[k][1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in out.items()}


def main():
    tag, trace, fetch, write = sys.argv[1:5]
    local_qubits = int(sys.argv[5]) if len(sys.argv) > 5 else 28
    prof = ROOT / "profiles"
    prof.mkdir(exist_ok=True)
    stats = glob.glob(f"{trace}/**/*_kernel_stats.csv", recursive=True)[0]
    shutil.copy(stats, prof / f"{tag}_kernel_stats.csv")
    dur = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(stats)):
        k = klass(r["Name"])
        if k:
            dur[k][0] += int(r["Calls"])
            dur[k][1] += float(r["TotalDurationNs"])
    fs, ws = counter_avg(fetch, "FETCH_SIZE"), counter_avg(write, "WRITE_SIZE")
    rows = []
    for k in sorted(dur):
        n, total = dur[k]
        row = {"kernel": k, "launches_in_stats_pass": n, "avg_duration_ms": total / n / 1e6}
        if k in fs and k in ws:
            row["fetch_size_KiB_avg"] = fs[k][1]
            row["write_size_KiB_avg"] = ws[k][1]
            row["hbm_bytes_per_launch"] = (2.0 * fs[k][1] + ws[k][1]) * 1024.0
        rows.append(row)
    doc = {"tag": tag, "csrc_sha16": source_hash(), "local_qubits": local_qubits, "source": "rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE "
                                 "(separate passes) on `python3 bench.py`",
           "correction": "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE = 1/2 of "
                         "wide coalesced reads; WRITE_SIZE exact for 16-B streaming stores)",
           "kernels": rows}
    (prof / f"{tag}_pmc_summary.json").write_text(json.dumps(doc, indent=1))
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
