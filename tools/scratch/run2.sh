set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_b
for sg in 0 4 12; do
echo "=== stagger $sg"
AVDF_PW_STAGGER=$sg timeout -k 10 600 python tools/pwgemm_bench.py > gpurun_out/r03_b/pw_fwd4_s$sg.log 2>&1 || { tail -30 gpurun_out/r03_b/pw_fwd4_s$sg.log; exit 1; }
grep -A1 "^fc1 768\|^fc2 768\|^proj 768\|^qkv 768\|^fc1 384" gpurun_out/r03_b/pw_fwd4_s$sg.log | grep -v "^--"
done
