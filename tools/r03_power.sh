#!/bin/bash
# Evidence for DESIGN.md's power note: the bare fp16 MFMA stream on zero vs random operands, with the shader clock measured
# in-kernel and rocm-smi's power / sclk sampled beside it.  Output: gpurun_out/$1/power.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r3_power}; mkdir -p $O
cd $R/tools/ubench && hipcc -O3 --offload-arch=gfx950 mfma_power.hip -o /tmp/mfma_power 2> $O/build.log || { cat $O/build.log; exit 1; }
( while true; do echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk' | tr -s ' ' | tr '\n' '|')"; sleep 0.4; done ) > $O/smi.log 2>&1 &
SMI=$!
sleep 1
echo "start $(date +%s.%N | cut -c1-14)" > $O/power.txt
timeout 120 /tmp/mfma_power 4 >> $O/power.txt 2>&1
echo "end $(date +%s.%N | cut -c1-14)" >> $O/power.txt
sleep 1
kill $SMI
echo "--- rocm-smi samples (0.4 s apart; the first 4 s of the run are the zero operands, the next 4 s the random ones)" >> $O/power.txt
cat $O/smi.log >> $O/power.txt
cat $O/power.txt | cut -c1-260
