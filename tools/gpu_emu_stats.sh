#!/bin/bash
# rocprofv3 kernel stats of the emulated N-rank factored step (compute side of one rank): which launches grow with N
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/emu
for n in ${@:-1 8}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/emu/n$n -o emu -- python3 bench.py --emulate-ranks $n --steps 100 --warmup 20 > gpurun_out/emu/n$n.json 2> gpurun_out/emu/n$n.err || exit 1
  f=$(find gpurun_out/emu/n$n -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/emu/n${n}_kernel_stats.csv
  echo "== N=$n"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
