# Development (GPU box): tiles per workgroup of k_extract_bp at 5 workgroups per CU
tools/ab.sh "r5t3 r5t3:FMD_BP_NT=4 r5t3:FMD_BP_NT=2" 2
tools/ab.sh "r5t3 r5t3:FMD_BP_NT=8 r5t3:FMD_BP_NT=2" 2 "--channels 1024"
tools/ab.sh "r5t3 r5t3:FMD_BP_NT=8 r5t3:FMD_BP_NT=4" 2 "--channels 2048"
