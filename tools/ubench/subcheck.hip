#include "field_secp256k1.h"
#include <cstdio>
#include <vector>
using namespace ecfft; using F = Secp256k1;
__global__ void k(const Fe256* a, const Fe256* b, Fe256* o, int n) { int i = threadIdx.x + blockIdx.x*blockDim.x; if (i<n) o[i] = F::sub(a[i], b[i]); }
int main(){ const int n=1<<16; std::vector<Fe256> a(n),b(n),o(n); uint64_t s=12345; auto rnd=[&](){s^=s<<13;s^=s>>7;s^=s<<17;return (uint32_t)(s>>16);};
 for(int i=0;i<n;++i){for(int l=0;l<8;++l){a[i].l[l]=rnd();b[i].l[l]=rnd();} a[i].l[7]&=0x7fffffff; b[i].l[7]&=0x7fffffff; if(i%7==0) b[i]=a[i]; if(i%11==0){ for(int l=1;l<8;++l) b[i].l[l]=a[i].l[l]; b[i].l[0]=a[i].l[0]+1; } if (i%13==0) a[i]=F::zero(); if(i%17==0){ for(int l=0;l<8;++l) b[i].l[l]=0xffffffff; b[i].l[0]=0xfffffc2e; b[i].l[1]=0xfffffffe; } }
 Fe256 *da,*db,*dd; hipMalloc(&da,n*32);hipMalloc(&db,n*32);hipMalloc(&dd,n*32); hipMemcpy(da,a.data(),n*32,hipMemcpyHostToDevice);hipMemcpy(db,b.data(),n*32,hipMemcpyHostToDevice);
 k<<<n/256,256>>>(da,db,dd,n); hipMemcpy(o.data(),dd,n*32,hipMemcpyDeviceToHost); int bad=0; for(int i=0;i<n;++i){Fe256 r=F::sub(a[i],b[i]); if(!F::eq(r,o[i])) ++bad;} printf("sub asm vs host: %d mismatches of %d\n",bad,n); return bad!=0; }
