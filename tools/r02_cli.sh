cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys, subprocess, time, os
sys.path.insert(0,'.')
from rabbitkssd_amd import synth
n,h,o=synth.clade_sketches(10000,1220,28)
synth.write_sketch_file('/tmp/b.sketch',10,6,3,n,h,o)
open('/tmp/b.sketch.dict','w').close(); open('/tmp/b.sketch.index','w').close()
tool=os.path.join(os.getcwd(),'rabbitkssd_amd','rabbit_kssd')
def run(env_extra, label):
    env=dict(os.environ, RK_TIMING='1', **env_extra)
    t0=time.time(); p=subprocess.run([tool,'alldist','-i','/tmp/b.sketch','-D','0.05','-o','o.out'],cwd='/tmp',env=env,capture_output=True); dt=time.time()-t0
    print('---', label, 'wall %.1f ms rc %d' % (dt*1e3, p.returncode))
    for l in p.stderr.decode().split('\n'):
        if 'timing' in l: print('   ', l)
for i in range(3): run({}, 'default')
run({'HIP_ENABLE_DEFERRED_LOADING':'0'}, 'deferred loading off')
run({'ROCR_VISIBLE_DEVICES':'0'}, 'ROCR_VISIBLE_DEVICES=0')
run({'HSA_ENABLE_SDMA':'0'}, 'sdma off')
run({'GPU_MAX_HW_QUEUES':'2'}, 'hw queues 2')
open('/tmp/h.cpp','w').write(r'''
#include <hip/hip_runtime.h>
#include <cstdio>
#include <sys/time.h>
static double now(){timeval tv;gettimeofday(&tv,0);return tv.tv_sec*1e3+tv.tv_usec/1e3;}
int main(){double t0=now();int n=0;(void)hipGetDeviceCount(&n);double t1=now();(void)hipSetDevice(0);void*p;(void)hipMalloc(&p,1<<20);double t2=now();hipStream_t s;(void)hipStreamCreateWithFlags(&s,hipStreamNonBlocking);double t3=now();void*h;(void)hipHostMalloc(&h,65536,0);double t4=now();printf("count %.1f ms, setdevice+malloc %.1f ms, stream %.1f ms, hostmalloc %.1f (%d devices)\n",t1-t0,t2-t1,t3-t2,t4-t3,n);return 0;}
''')
subprocess.run(['/opt/rocm/bin/hipcc','-O2','/tmp/h.cpp','-o','/tmp/h'],check=True,capture_output=True)
for i in range(2):
    t0=time.time(); p=subprocess.run(['/tmp/h'],capture_output=True); print('trivial hip program wall %.1f ms:'%((time.time()-t0)*1e3), p.stdout.decode().strip())
t0=time.time(); subprocess.run(['/bin/true']); print('exec /bin/true %.1f ms'%((time.time()-t0)*1e3))
PY
