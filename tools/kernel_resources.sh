#!/bin/bash
# Register, scratch and LDS usage of every kernel of the library as the compiler reports it (-Rpass-analysis=kernel-resource-usage),
# plus the scratch stores / loads of the headline variant of k_trace and where its stores sit (prologue or main loop).  CPU only.
# usage: tools/kernel_resources.sh > profiles/<tag>_kernel_resources.txt
set -o pipefail
here=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
cd $here/raytracing_amd/csrc
flags=$(make -s -p -n 2>/dev/null | sed -n 's/^FLAGS *:\?= *//p' | head -1)
[ -z "$flags" ] && flags="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -I$here/include"
echo "# hipcc $flags -Rpass-analysis=kernel-resource-usage, one line per kernel: VGPRs, AGPRs, SGPRs, spilled registers, scratch bytes per lane, occupancy (waves per SIMD), LDS bytes (static)"
for f in rt_trace.hip rt_wavefront.hip rt_pixel.hip rt_ppm.hip rt_scene.hip rt_unit.hip; do
  /opt/rocm/bin/hipcc $flags -save-temps=obj -Rpass-analysis=kernel-resource-usage -o $tmp/${f%.hip}.so $f 2> $tmp/${f%.hip}.log
  python3 - $tmp/${f%.hip}.log $f <<'PY'
import re, subprocess, sys
rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"remark: [^:]+:\d+:\d+: +(.*?) \[-Rpass-analysis", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    print("%-14s %-44s VGPRs %3s AGPRs %3s SGPRs %3s spilled VGPRs %3s SGPRs %3s scratch %4s B/lane  waves/SIMD %s  LDS %s B" % (sys.argv[2], name, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
PY
done
s=$tmp/rt_trace-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^_Z7k_traceILb1ELb1ELb0ELb1ELi1ELb0EEv9TraceArgs:/,/s_endpgm/' $s > $tmp/kt.s
first_atomic=$(grep -n "global_atomic_add " $tmp/kt.s | head -1 | cut -d: -f1)
echo "# k_trace<lds, plane, no hierarchy, matrix-pipe filter, rows from k_primary's queue> (the headline variant): $(wc -l < $tmp/kt.s) lines of ISA; the main loop's first hand-out atomic is at line ${first_atomic:-?}, everything before line ~400 is the prologue"
echo "#   scratch stores at lines: $(grep -n scratch_store $tmp/kt.s | cut -d: -f1 | tr '\n' ' ')"
echo "#   scratch loads: $(grep -c scratch_load $tmp/kt.s) (reloads of loop-invariant values; the stores all sit in the prologue when every line number above is below the hand-out atomic's)"
echo "#   v_mfma instructions: $(grep -c v_mfma $tmp/kt.s), s_setprio: $(grep -c s_setprio $tmp/kt.s)"
rm -rf $tmp
