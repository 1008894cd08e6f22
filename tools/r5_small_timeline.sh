# Development (GPU box): kernel timelines of the small-batch workloads (wideband configs[4], 1024 stations) — what the step is made of
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for a in "--wideband" "--channels 1024"; do
  rm -rf /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py --no-cpu-baseline --no-other-mode --no-configs --no-host-fed --no-kernel-times $a > /tmp/tr.json 2>/tmp/tr.err
  echo "== $a"; tail -c 400 /tmp/tr.json; echo
  python3 tools/timeline.py /tmp/tr 24
done
