"""Build a side variant of the kernel library from the working tree with text substitutions applied to csrc/attention.hip (or MKVARIANT_FILE=gemm.hip ...) (timing-only ablations and
A/B candidates; run with ULLSAM_HIP_LIB or tools/attn_lib_ab.py): python tools/probes/mkvariant.py <name> "old=>new" ...  -> ullsam_amd/lib/libullsam_hip_<name>.so"""
import sys, subprocess, os, shutil, tempfile
# usage: mkvariant.py name  "old1=>new1" ...  builds ullsam_amd/lib/libullsam_hip_<name>.so from the working tree with text substitutions in attention.hip
ROOT='/root/repo'
sys.path.insert(0, ROOT)
from ullsam_amd import build as B
name=sys.argv[1]
FILE=os.environ.get('MKVARIANT_FILE','attention.hip')   # which csrc file the substitutions apply to
src=open(f'{ROOT}/ullsam_amd/csrc/{FILE}').read()
for sub in sys.argv[2:]:
    a,b=sub.split('=>')
    assert a in src, a
    src=src.replace(a,b)
td=tempfile.mkdtemp()
shutil.copytree(f'{ROOT}/ullsam_amd/csrc', f'{td}/csrc')
open(f'{td}/csrc/{FILE}','w').write(src)
obj=f'{td}/variant.o'
r=subprocess.run([B.HIPCC,*B.FLAGS,'-c',f'{td}/csrc/{FILE}','-o',obj],capture_output=True,text=True)
assert r.returncode==0, r.stderr[-2000:]
objs=[os.path.join(B.LIBDIR,'obj',f) for f in os.listdir(os.path.join(B.LIBDIR,'obj')) if f.endswith('.o') and f!=FILE.replace('.hip','.o')]+[obj]
out=os.path.join(B.LIBDIR,f'libullsam_hip_{name}.so')
subprocess.run([B.HIPCC,'--offload-arch=gfx950','-shared','-fPIC','-o',out,*objs],check=True)
print(out)
