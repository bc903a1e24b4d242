import sys
p=2**256-2**32-977
R=2**29
ok=True;n=0
for line in sys.stdin:
    if not line.startswith("CHK"):
        print(line.rstrip()); continue
    v=list(map(int,line.split()[1:]))
    f=lambda a:sum(x*R**i for i,x in enumerate(a))
    t,u,x,c,o=(v[9*i:9*i+9] for i in range(5))
    xl=f(x[:5]); xh=f(x[5:])
    want=(f(t)*xl+f(u)*xh+f(c))%p
    got=f(o)%p
    n+=1
    if want!=got or max(o)>=2**30: ok=False; print("MISMATCH",o)
print("exactness:", "OK" if ok else "FAIL", n, "lanes")
