# tools/h2d_probe.sh -- GPU-box helper: builds and runs tools/probe/h2d_probe.hip
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -o /tmp/h2d_probe tools/probe/h2d_probe.hip || exit 1
/tmp/h2d_probe 128
