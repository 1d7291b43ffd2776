"""Where the load time of `fmsi query --tables X` goes at human scale (FMSI_HIP_VERBOSE phases of the CLI process):
    python tools/cli_load_time.py [compact|full|auto]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fmsi_amd as fm  # noqa: E402

tables = sys.argv[1] if len(sys.argv) > 1 else "compact"
d = "/dev/shm/fmsi_cli_load"
os.makedirs(d, exist_ok=True)
prefix = os.path.join(d, "idx")
if not os.path.exists(prefix + ".fmsi.misc"):
    fm.lib()
    dev = fm.DeviceIndex.build_synthetic(2_800_000_000, 1234, 31)
    dev.write_files(prefix)
    dev.close()
q = os.path.join(d, "q.fa")
with open(q, "w") as f:
    f.write(">r\n" + "ACGTTGCATGGCATCGATCGATCGGCTAGCTAGCATCGATCGACTAGCTAGCTAGCTAGCATCGAC" * 20 + "\n")
env = dict(os.environ, FMSI_HIP_VERBOSE="1")
r = subprocess.run([str(fm.CLI_PATH) if hasattr(fm, "CLI_PATH") else os.path.join(os.path.dirname(fm.__file__), "bin", "fmsi"), "query", "--tables", tables, "--stats", "-k", "31", "-q", q, prefix],
                   env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
print(r.stderr.decode()[-6000:])
