import numpy as np
b1,b2,eps=0.9,0.999,1e-7
def lr_t(lr,t): return lr*np.sqrt(1-b2**t)/(1-b1**t)
def exact(th,m,v,last,n,lr,dtype=np.float64):
    th,m,v=dtype(th),dtype(m),dtype(v)
    for k in range(1,n+1):
        m=dtype(m*dtype(b1)); v=dtype(v*dtype(b2))
        th=dtype(th-dtype(dtype(lr_t(lr,last+k))*m)/dtype(np.sqrt(v)+dtype(eps)))
    return th,m,v
def series(th,m,v,last,n,lr,order=3):
    f=np.float32
    r=f(np.sqrt(np.float64(f(b2)))); omr=f(1-np.sqrt(np.float64(f(b2))))
    T0=Z1=Z2=Z3=f(0); p1=f(1); p2=f(1); z=f(0)
    for k in range(1,min(n,1024)+1):
        p1=f(p1*f(b1)); p2=f(p2*f(b2)); z=f(z*r+omr)
        w=f(f(lr_t(lr,last+k))*p1)
        T0=f(T0+w); wz=f(w*z); Z1=f(Z1+wz); wz=f(wz*z); Z2=f(Z2+wz); wz=f(wz*z); Z3=f(Z3+wz)
    if n>1024: p1=f(0); p2=f(np.float64(f(b2))**n)
    a0=f(np.sqrt(f(v))); den=f(a0+f(eps)); inv=f(f(1)/den); u=f(a0*inv)
    if order==3: S=f(T0+u*f(Z1+u*f(Z2+u*Z3)))
    elif order==2: S=f(T0+u*f(Z1+u*Z2))
    else: S=f(T0+u*Z1)
    return f(f(th)-f(f(m)*inv)*S), f(f(m)*p1), f(f(v)*p2)
rs=np.random.RandomState(0)
worst=0
for trial in range(3000):
    g=10**rs.uniform(-9,0)*rs.choice([-1,1]); cnt=rs.randint(1,5)
    m=0.1*g*cnt*rs.uniform(0.3,1); v=0.001*g*g*cnt*rs.uniform(0.3,1)
    if rs.rand()<0.1: v=10**rs.uniform(-30,-14)
    th=rs.uniform(-0.1,0.1); last=rs.randint(1,3000); n=rs.choice([1,2,5,10,30,100,400,2000])
    lr=0.001
    e=exact(th,m,v,last,n,lr); e32=exact(th,m,v,last,n,lr,np.float32)
    for order in (1,2,3):
        s=series(th,m,v,last,n,lr,order)
        err=abs(float(s[0])-e[0]); 
        if order==3:
            worst=max(worst,err)
            if err>2e-7: print('bad',trial,g,m,v,n,err,abs(float(e32[0])-e[0]), abs(e[0]-th))
    if trial<12: print(n, 'move',abs(e[0]-th),'err o1',abs(float(series(th,m,v,last,n,lr,1)[0])-e[0]),'o2',abs(float(series(th,m,v,last,n,lr,2)[0])-e[0]),'o3',abs(float(series(th,m,v,last,n,lr,3)[0])-e[0]),'fp32 stepwise',abs(float(e32[0])-e[0]), 'm rel',abs(float(s[1])/e[1]-1),'v rel',abs(float(s[2])/e[2]-1))
print('worst o3',worst)
