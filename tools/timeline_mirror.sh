#!/bin/bash
# development: kernel timeline of ONE receiver behind the drop-in classes (tests/cpp/mirror_harness in its timing mode, 18 sub-channels):
# shows stream A (synchroniser, demodulation, phase tail) of frame k + 1 running beside stream B's decode of frame k
#   gpurun -- 'bash tools/timeline_mirror.sh > gpurun_out/timeline_mirror.txt'
export TMPDIR=/tmp
D=gpurun_out/tm; rm -rf $D; mkdir -p $D
python3 - <<PY
import sys, os, numpy as np, torch
sys.path[:0] = ["dab-radio_amd", "tools"]
import dabgpu, dabsynth
dev = torch.device("cuda", 0)
prs, mapper, _ = dabgpu.host_tables()
mux = dabsynth.Multiplex(1, 21, dev)
f2 = dabsynth.modulate(mux.frame_bits[0], prs, mapper)
n_frames = 60
x = torch.cat([torch.zeros(5000 + 2656, dtype=torch.complex64, device=dev), f2.reshape(-1).repeat(n_frames // 2)])
n = torch.arange(x.numel(), device=dev, dtype=torch.float64)
x = x * torch.polar(torch.ones_like(n), 2 * np.pi * 1.3e-3 * n).to(torch.complex64)
x[:5000] = x[-5000:]
x = x + 0.02 * torch.view_as_complex(torch.randn((x.numel(), 2), device=dev))
x.cpu().numpy().astype(np.complex64).tofile("$D/iq.c32")
PY
ARGS=""; for s in $(seq 0 17); do ARGS="$ARGS $((48*s)) 48 2 0"; done
export DABGPU_HARNESS_BENCH=1 LD_LIBRARY_PATH=dab-radio_amd:/opt/rocm/lib:$LD_LIBRARY_PATH
rocprofv3 --kernel-trace --output-format csv -d $D/prof -o t -- ./tests/cpp/mirror_harness $D/iq.c32 $D 65536 $ARGS > $D/stdout.log 2>&1
tail -1 $D/stdout.log
f=$(find $D/prof -name '*kernel_trace.csv' | head -1)
n=$(python3 -c "
import csv
rows=[r for r in csv.DictReader(open('$f')) if 'dabgpu' in r['Kernel_Name']]
print(len(rows))")
echo "dabgpu kernel launches: $n"
python3 tools/ktimeline.py $f $((n-150)) 60
rm -rf $D
