#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_pcnn; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
cat > /tmp/pc.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
from vqvae_amd.pixelcnn import GatedPixelCNN
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = GatedPixelCNN(512, 64, 15, 10).to(dev).eval()
x = torch.randint(0, 512, (1024, 8, 8), device=dev); label = torch.randint(0, 10, (1024,), device=dev)
with torch.no_grad():
    for _ in range(8): m(x, label)
torch.cuda.synchronize()
PY
(cd $R && timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof -- python /tmp/pc.py > $O/prof.log 2>&1)
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB 20 > $O/kernel_stats.txt
rm -rf $O/prof
cut -c1-170 $O/kernel_stats.txt
