# VERDICT r03 item 4, measured before it is built: what would re-associating the polynomial chains (Estrin instead of Horner) buy a lone
# path's bounce?  libbeifong_hip_estrin.so = make variant VARIANT=estrin EXTRA=-DBF_ESTRIN_PROBE (a TIMING probe: its results differ
# from the oracle's in the last bits; nothing compares them).  Isolated renders (one render alone on the GPU, its own tail) and the
# pipelined step, product build vs probe, same box.
cd $GRAFT_REPO_ROOT
bash tools/r04_ab.sh r04_estrin new estrin -- c3 c4shard c2
