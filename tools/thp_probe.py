import numpy as np, time, ctypes
libc=ctypes.CDLL(None)
n=4<<30
for adv in (False, True, False, True):
    a=np.empty(n+(2<<20),dtype=np.uint8)
    addr=a.ctypes.data; al=(addr+(2<<20)-1)&~((2<<20)-1)
    r=0
    if adv: r=libc.madvise(ctypes.c_void_p(al), ctypes.c_size_t(n), 14)
    t=time.time(); a[al-addr:al-addr+n:4096]=1; dt=time.time()-t
    print('madvise' if adv else 'plain  ', r, 'first touch of 4 GiB: %.3f s'%dt, flush=True)
    del a
print(open('/sys/kernel/mm/transparent_hugepage/enabled').read())
