import sys,re,subprocess
txt=subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf","--notes",sys.argv[1]],capture_output=True,text=True).stdout
blocks=txt.split('- .agpr_count')
for b in blocks[1:]:
    nm=re.search(r'\.name:\s+(\S+)',b); vg=re.search(r'\.vgpr_count:\s+(\d+)',b); sg=re.search(r'\.sgpr_count:\s+(\d+)',b); sc=re.search(r'\.private_segment_fixed_size:\s+(\d+)',b); lds=re.search(r'\.group_segment_fixed_size:\s+(\d+)',b)
    if nm and any(k in nm.group(1) for k in sys.argv[2:]):
        dem=subprocess.run(["c++filt",nm.group(1)],capture_output=True,text=True).stdout.strip()
        print(dem[:90], 'vgpr',vg.group(1),'sgpr',sg.group(1),'scratch',sc.group(1),'lds',lds.group(1))
