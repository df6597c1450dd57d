# Per-process spread of the HBM-resident step: N processes, each creates ONE context of the workload, prints where its fields were
# allocated (TRM_DEBUG_PLACEMENT) and times the step.  usage: bash profiles/tools/placement_probe.sh [workload] [processes]
WL=${1:-c3x8}; N=${2:-12}
L=gpurun_out/r04_placement_probe_$WL.log; : > $L
for i in $(seq 1 $N); do
  TRM_DEBUG_PLACEMENT=1 timeout -k 10 300 python profiles/tools/ab_options.py $WL p$i: --steps 60 --reps 5 >> $L 2>&1 || exit 1
done
python - $L <<'PY'
import sys, json, re
fields, out = {}, []
for line in open(sys.argv[1]):
    m = re.match(r"trm placement: field (\d+) raw (0x[0-9a-f]+) bytes (\d+)", line)
    if m:
        fields.setdefault(int(m.group(1)), int(m.group(2), 16))
    elif line.startswith("{"):
        d = json.loads(line)
        (name, v), = d["us_per_step"].items()
        out.append((v["median"], name, dict(fields)))
        fields = {}
for t, name, f in sorted(out):
    print(f"{t:8.2f} {name:4s} " + " ".join(f"{k}:{(v >> 21) & 0xfff:03x}" for k, v in sorted(f.items())[:8]))
PY
