#!/bin/bash
# Round-5 experiments on sk_runs_kernel (profiles/r05_sk_runs_isa_sections.txt) and the ORD flush, on the GPU box: variants are built there.
#   gpurun -- bash tools/exp_r05_sk.sh
cd "$(dirname "$0")/.."
leg() { python bench.py --leg "$1" --warmup 3 --no-pmc 2>/dev/null | tail -1 | python -c "
import json, sys
for l in json.loads(sys.stdin.read()):
    print('   ', l['name'].ljust(28), round(l['value'] / 1e9, 2), 'G/s  ms', l['ms'], 'kernel ms', l.get('avg_kernel_ms'), 'requests/kmer', l.get('requests_per_kmer'), 'ok' if l['matches_reference'] else 'MISMATCH')"; }
echo "== baseline c4sk"; leg c4sk
echo "== ORD at t = 16: kernel writes the -1s"; FMSI_BENCH_KLCP_TABLE=16 leg c4k
echo "== ORD at t = 16: with round 4's pre-fill pass as well"; FMSI_HIP_STREAM_ORD_PREFILL=1 FMSI_BENCH_KLCP_TABLE=16 leg c4k
cp fmsi_amd/csrc/skdict.inc /tmp/skdict.inc.orig; cp fmsi_amd/csrc/fmsi_dict.hip /tmp/fmsi_dict.hip.orig
sed -i 's/kPend = 4u/kPend = 6u/' fmsi_amd/csrc/skdict.inc
make -C fmsi_amd/csrc -j48 > /dev/null 2>&1 && { echo "== kPend = 6 (a flush per 6 groups; 14 waves per CU)"; leg c4sk; }
cp /tmp/skdict.inc.orig fmsi_amd/csrc/skdict.inc
sed -i 's/sk_mix(sk_mix(xc, mbits, kSkScoreSalt), mbits, kSkBucketSalt)/sk_mix(xc, mbits, kSkBucketSalt)/' fmsi_amd/csrc/skdict.inc fmsi_amd/csrc/fmsi_dict.hip
sed -i 's/sk_mix(sk_mix(flipped ? y : x, 2u \* static_cast<uint32_t>(m), kSkScoreSalt), 2u \* static_cast<uint32_t>(m), kSkBucketSalt)/sk_mix(flipped ? y : x, 2u * static_cast<uint32_t>(m), kSkBucketSalt)/' fmsi_amd/csrc/skdict.inc
grep -c "sk_mix(sk_mix" fmsi_amd/csrc/skdict.inc fmsi_amd/csrc/fmsi_dict.hip
make -C fmsi_amd/csrc -j48 > /dev/null 2>&1 && { echo "== one sk_mix instead of two"; leg c4sk; leg c3sk; }
cp /tmp/skdict.inc.orig fmsi_amd/csrc/skdict.inc; cp /tmp/fmsi_dict.hip.orig fmsi_amd/csrc/fmsi_dict.hip
