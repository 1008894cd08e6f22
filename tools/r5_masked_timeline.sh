cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp fm-radio_amd/csrc/libfmdemod.so /tmp/orig.so; cp tools/ab/r5base.so fm-radio_amd/csrc/libfmdemod.so
export FMD_CU_MASK_F=ffffffff-ffffffff-ffffffff-0000ffff-00000000-00000000-00000000-00000000 FMD_CU_MASK_X=00000000-00000000-00000000-ffff0000-ffffffff-ffffffff-ffffffff-ffffffff
rm -rf /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py --no-cpu-baseline --no-other-mode --no-configs --no-host-fed --no-kernel-times > /tmp/tr.json 2>/tmp/tr.err
tail -c 300 /tmp/tr.json; echo
python3 tools/timeline.py /tmp/tr 30
cp /tmp/orig.so fm-radio_amd/csrc/libfmdemod.so
