cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/proto_sq; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/a -o pmc -- tools/micro/bin/x6_tile_proto 786432 8 1 > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $O/b -o pmc -- tools/micro/bin/x6_tile_proto 786432 8 1 > $O/b.log 2>&1
rm -f $O/*/pmc_kernel_trace.csv
ls $O/a $O/b; grep -E "^(P|B)" $O/a.log | head -5 | cut -c1-160
