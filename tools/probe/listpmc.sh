cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ|TA|TCP|TCC|TD|GRBM)_[A-Z0-9_]+" | sort -u > gpurun_out/pmc_list.txt
wc -l gpurun_out/pmc_list.txt
