"""Phase times of the super-k-mer dictionary's construction at human scale (FMSI_HIP_VERBOSE): python tools/sk_build_time.py [chain]"""
import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["FMSI_HIP_VERBOSE"] = "1"
import fmsi_amd as fm
fm.lib()
chain = len(sys.argv) > 1 and sys.argv[1] == "chain"
ctx = fm.builder_chain(0) if chain else None
if ctx:
    ctx.__enter__()
t0 = time.time()
dev = fm.DeviceIndex.build_synthetic(2_800_000_000, 1234, 31)
print("build", time.time() - t0, flush=True)
t0 = time.time()
dev.enable_dict(superkmer_only=True)
print("enable_dict(superkmer_only)", time.time() - t0, "scratch kept", fm.scratch_bytes() if hasattr(fm, "scratch_bytes") else None, flush=True)
if ctx:
    ctx.__exit__(None, None, None)
