#!/bin/bash
# rocprofv3 kernel stats of the sorted LF walk at human scale: gpurun -- bash tools/prof_sorted.sh r04d
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_sorted" -o run -- python3 "$R/tools/lf_sorted_probe.py" 2.8e9 134217728 only-sorted > "$O/sorted_probe.log" 2>&1
python3 - "$O" <<'P'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/trace_sorted/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(sys.argv[1] + "/sorted_kernel_stats.txt", "w") as out:
        for r in rows[:25]:
            line = f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e6:9.3f} ms total {float(r["TotalDurationNs"]) / 1e6:9.2f} ms'
            print(line); out.write(line + "\n")
P
python3 - "$O" <<'P'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/trace_sorted/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if any(x in r["Kernel_Name"] for x in ("sorted_walk", "sw_", "onesweep"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    with open(sys.argv[1] + "/sorted_dispatches.txt", "w") as out:
        for r in rows[-40:]:
            line = f'{r["Kernel_Name"][:70]:70s} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6:9.3f} ms  grid {r.get("Grid_Size_X", r.get("Grid_Size", ""))}'
            print(line); out.write(line + "\n")
P
find "$O" -name '*_kernel_trace.csv' -delete; find "$O" -name '*_agent_info.csv' -delete
tail -8 "$O/sorted_probe.log"
