# tools/exit_probe.sh -- GPU-box helper: wall time of tools/probe/exit_probe under several footprints / ways to leave
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -Wno-unused-result -lpthread -o /tmp/exit_probe tools/probe/exit_probe.hip || exit 1
python3 - <<'PY'
import subprocess, time
for cfg in ("0 0 0 exit", "26 320 3 exit", "26 320 3 exit 1", "26 320 3 exit 4", "26 320 3 exit 8", "26 320 3 exit 4 30", "0 0 0 exit 4", "0 0 0 exit 0 30"):
    for rep in range(2):
        t = time.perf_counter()
        pr = subprocess.run(["/tmp/exit_probe"] + cfg.split(), capture_output=True, text=True)
        print(f"vram_gb pin_mb host_gb how = {cfg:22s}: wall {time.perf_counter() - t:.3f} s ; {pr.stderr.strip()}", flush=True)
PY
