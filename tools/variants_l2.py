import os, subprocess, sys
sys.path.insert(0, '/root/repo')
from voicebridge_amd import build as B
name, flags = sys.argv[1], sys.argv[2:]
vdir = os.path.join(B.LIBDIR, 'variants'); os.makedirs(vdir, exist_ok=True)
obj = os.path.join(vdir, name + '_l2.o')
subprocess.check_call(['/opt/rocm/bin/hipcc'] + B.FLAGS + ['-ffp-contract=off'] + B.LAYOUT2_FLAGS + flags + ['-c', os.path.join(B.CSRC, 'vbh_decode.hip'), '-o', obj])
objs = [os.path.join(B.LIBDIR, 'obj', s.replace('.hip', '.o')) for s in B.HIP_SOURCES] + [obj]
lib = os.path.join(vdir, 'libvoicebridge_hip_%s.so' % name)
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
print('built', lib)
