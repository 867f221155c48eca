# runs ON THE GPU BOX: kernel times vs the number of workgroups the persistent kernels get (-DBF_EXPERIMENT build as variants/exp.so)
cd $GRAFT_REPO_ROOT
export FASTNERF_LIB=$PWD/fast-learning-nerf_amd/variants/exp.so
for w in 512 480 448 416 384 320 256; do echo -n "fwd_wgs $w: "; FASTNERF_FWD_WGS=$w python tools/time_mlp.py 2>/dev/null | tail -1; done
for w in 512 448 384; do FASTNERF_DX_WGS=$w python tools/time_bwd_parts.py; done
