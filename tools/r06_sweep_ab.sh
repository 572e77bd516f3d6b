#!/bin/bash
# per-kernel times of the streamed-codebook quantizer (tools/r06_sweep_ko.py) for several builds of the library:  r06_sweep_ab.sh lib1.so [lib2.so ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for l in "$@"; do
  O=/tmp/sweep_ab_$(basename $l .so); rm -rf $O
  (cd $R && VQVAE_HIP_LIB_OVERRIDE=$([ "$l" = base ] && echo "" || echo $l) timeout 200 rocprofv3 --kernel-trace --stats -d $O -- python tools/r06_sweep_ko.py > $O.log 2>&1)
  DB=$(find $O -name "*.db" | head -1)
  echo "== $l"; python $R/tools/rocprof_summary.py $DB 6 | grep -E "sweep|resolve|rows16|gather" | cut -c1-150
done
