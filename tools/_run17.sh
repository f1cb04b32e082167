R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02q; mkdir -p $O; cd $R
timeout 120 tools/ubench/aes_forms > $O/aes_forms.txt 2>&1; cat $O/aes_forms.txt
echo "== production, 1024 instances, two per workgroup" > $O/ni4.txt
KAB_CT_CAP=1 timeout 900 python3 tools/kernel_ab.py 1024 >> $O/ni4.txt 2>&1
echo "== experiment, 1024 instances, FOUR per workgroup (quarter window)" >> $O/ni4.txt
GSV_LDS_SLOTS_CAP=1440 KAB_CT_CAP=1 GSV_ENGINE_SO=$R/garbled_snark_verifier_amd/libgsv_engine_n4.so timeout 900 python3 tools/kernel_ab.py 1024 >> $O/ni4.txt 2>&1
echo "== production, 1024 instances, two per workgroup, quarter window (what the smaller window alone costs)" >> $O/ni4.txt
GSV_LDS_SLOTS_CAP=1440 KAB_CT_CAP=1 KAB_NOCHECK=1 timeout 900 python3 tools/kernel_ab.py 1024 >> $O/ni4.txt 2>&1
cat $O/ni4.txt
