cd $GRAFT_REPO_ROOT
Q=tests/golden/rand_sars_k13.fa
for i in 1 2 3; do /usr/bin/time -f "wall %e s" fmsi_amd/bin/fmsi query -q $Q tests/golden/idx/sars_k13 > /dev/null; done
FMSI_HIP_VERBOSE=1 fmsi_amd/bin/fmsi query --stats -q $Q tests/golden/idx/sars_k13 2>&1 >/dev/null | tail -12
for i in 1 2 3; do /usr/bin/time -f "ref wall %e s" oracle/_ref/fmsi query -q $Q tests/golden/idx/sars_k13 > /dev/null; done
python3 - <<'PY'
import subprocess, time
t=time.time()
for i in range(20):
    subprocess.run(["fmsi_amd/bin/fmsi","query","-q","tests/golden/rand_sars_k13.fa","tests/golden/idx/sars_k13"],capture_output=True)
print("20 runs via subprocess: %.3f s each" % ((time.time()-t)/20))
t=time.time()
for i in range(20):
    subprocess.run(["fmsi_amd/bin/fmsi","query","-S","-O","-q","tests/golden/rand_sars_k13.fa","tests/golden/idx/sars_k13"],capture_output=True)
print("20 runs -S -O via subprocess: %.3f s each" % ((time.time()-t)/20))
PY
