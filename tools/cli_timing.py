#!/usr/bin/env python3
"""Developer tool (GPU box): where does the wall time of `rabbit_kssd alldist` go?  Writes the bench's 10,000-sketch
file, runs the tool with RK_TIMING=1 a few times and prints its stamps; for comparison a HIP program that only
initialises the runtime and /bin/true."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rabbitkssd_amd import synth  # noqa: E402


def main(runs=4):
    n, h, o = synth.clade_sketches(10000, 1220, 28)
    synth.write_sketch_file('/tmp/b.sketch', 10, 6, 3, n, h, o)
    for f in ('/tmp/b.sketch.dict', '/tmp/b.sketch.index'):
        open(f, 'w').close()
    tool = os.path.join(ROOT, 'rabbitkssd_amd', 'rabbit_kssd')
    for i in range(runs):
        env = dict(os.environ, RK_TIMING='1')
        t0 = time.time()
        p = subprocess.run([tool, 'alldist', '-i', '/tmp/b.sketch', '-D', '0.05', '-o', 'o.out'], cwd='/tmp', env=env, capture_output=True)
        print('--- run %d: wall %.1f ms rc %d' % (i, (time.time() - t0) * 1e3, p.returncode))
        for line in p.stderr.decode().split('\n'):
            if 'timing' in line:
                print('   ', line)
    src = '/tmp/hipinit.cpp'
    open(src, 'w').write('#include <hip/hip_runtime.h>\n#include <cstdio>\nint main(){int n=0;(void)hipGetDeviceCount(&n);(void)hipSetDevice(0);void*p;(void)hipMalloc(&p,1<<20);printf("%d devices\\n",n);return 0;}\n')
    subprocess.run(['/opt/rocm/bin/hipcc', '-O2', src, '-o', '/tmp/hipinit'], check=True, capture_output=True)
    for i in range(3):
        t0 = time.time()
        subprocess.run(['/tmp/hipinit'], capture_output=True)
        print('HIP program that only initialises the runtime: wall %.1f ms' % ((time.time() - t0) * 1e3))
    t0 = time.time()
    subprocess.run(['/bin/true'])
    print('/bin/true: %.1f ms' % ((time.time() - t0) * 1e3))


if __name__ == '__main__':
    main()
