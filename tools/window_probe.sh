# tools/window_probe.sh -- GPU-box helper: builds and runs tools/probe/window_probe.hip (the 2-bit reference image question)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -o /tmp/window_probe tools/probe/window_probe.hip || exit 1
for L in 150 55; do echo "# read length $L"; /tmp/window_probe 14285714 $L || exit 1; done
