cd $GRAFT_REPO_ROOT
touch canonicalvoting_amd/csrc/sparse_conv.hip
CV_SC_DEFS="-DCV_WP_NPRE=27" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
python profiles/groups_micro.py 2>&1 | grep -v amdgpu.ids
