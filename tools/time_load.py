"""Times the stages of bringing an index from the reference's files into HBM with all accelerators (what `fmsi query`
does before its first batch): python tools/time_load.py [n].  Human scale on the gpurun box: host load 0.5 s, layout +
upload 0.5 s, t = 16 table 0.3 s, pair layout 0.03 s, candidate tables 0.9 s."""
import sys, time
sys.path.insert(0, '.')
import torch
import fmsi_amd as fm
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_800_000_000
d = fm.DeviceIndex.build_synthetic(n, 12345, 31, with_klcp=True)
t0 = time.time(); d.write_files('/dev/shm/tl_idx'); print('write_files', round(time.time() - t0, 3))
d.close(); del d
t0 = time.time(); h = fm.HostIndex('/dev/shm/tl_idx', use_klcp=False); t1 = time.time(); print('host load (files + RRR decode)', round(t1 - t0, 3))
h.view.prefix_len = 0
t0 = time.time(); dev = fm.DeviceIndex(h.view, 0, keepalive=h); print('create without table (host layout + H2D)', round(time.time() - t0, 3))
dev.close()
h.view.prefix_len = 16
t0 = time.time(); dev = fm.DeviceIndex(h.view, 0, keepalive=h); print('create with t=16 table', round(time.time() - t0, 3))
t0 = time.time(); dev.enable_pair_layout(); print('pair layout', round(time.time() - t0, 3))
t0 = time.time(); dev.enable_verify(); print('candidate tables', round(time.time() - t0, 3))
import os, glob
for f in glob.glob('/dev/shm/tl_idx*'): os.remove(f)
