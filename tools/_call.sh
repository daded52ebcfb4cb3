set -e
cd $GRAFT_REPO_ROOT
for v in "" .ab_libs/ma_h6.so .ab_libs/ma_h7.so .ab_libs/ma_h8.so .ab_libs/ma_h9.so "" .ab_libs/ma_h7.so; do
echo "--- lib '$v'" >> gpurun_out/hook.txt; I2V_LIB_PATH=$v timeout -k 10 300 python tools/attn_outproj_probe.py 131072 16 2>&1 | grep "pair" >> gpurun_out/hook.txt
done
