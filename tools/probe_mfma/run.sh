# build here (cross-compiles): hipcc --offload-arch=gfx950 -O3 -o tools/probe_mfma/mfma_shape_probe tools/probe_mfma/mfma_shape_probe.hip
# run on the GPU box:          tools/probe_mfma/mfma_shape_probe
