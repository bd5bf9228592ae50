# round 6, GPU call H5: the file-fed rate through pooled per-reader bounce buffers + non-temporal stores.   bash tools/gpu_r06h.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r06h5
rm -rf $O; mkdir -p $O
for kb in 1024 0 1024 0; do
  echo "## TWX_IO_BOUNCE_KB=$kb" >> $O/io_rate.txt
  TWX_IO_BOUNCE_KB=$kb timeout 120 python3 tools/io_rate.py 96 8 16 32 2>/dev/null | grep -v "^pinned\|host buffer" >> $O/io_rate.txt
done
echo "## host buffer + link" >> $O/io_rate.txt; timeout 120 python3 tools/io_rate.py 48 32 2>/dev/null | tail -2 >> $O/io_rate.txt
cat $O/io_rate.txt
