import sys, time
sys.path.insert(0, '/root/repo')
import fmsi_amd as fm
klcp = len(sys.argv) > 1 and sys.argv[1] == "klcp"
t0 = time.time()
idx = fm.DeviceIndex.build_synthetic(2_800_000_000, 12345, 31, with_klcp=klcp)
print("build", time.time() - t0, file=sys.stderr)
t0 = time.time(); idx.enable_pair_layout(); print("pair", time.time() - t0, file=sys.stderr)
t0 = time.time(); idx.enable_dict(); print("dict", time.time() - t0, file=sys.stderr)
