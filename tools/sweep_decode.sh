set -e
for tpq in 8 4; do for tgt in 512 2048; do
echo "tpq=$tpq target=$tgt"; NVX_DEC_NOSPLIT=100000 NVX_DEC_TPQ=$tpq NVX_DEC_TARGET=$tgt timeout -k 10 120 python -u tools/kernel_bench.py decode --every 16 > /tmp/o.txt 2>&1 || { tail -5 /tmp/o.txt; exit 1; }
grep -E "^0 |^256 |^384 |^512 |^640 |^704 |^768 |kernel" /tmp/o.txt
done; done
