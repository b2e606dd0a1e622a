// Host-side check of the unreduced edwards25519 mixed addition (csrc/ed25519.hpp xyzz_madd_lazy) against the eager unified
// addition over a random walk with both signs and the identity mixed in; coordinates must stay below 2p.
#include <cstdio>
#include <cstdlib>
#include "../../bulletproofsplus_amd/csrc/ristretto.hpp"
using namespace bpp;
static uint64_t st=0x1234567;
static uint64_t rnd(){ st+=0x9E3779B97F4A7C15ull; uint64_t z=st; z=(z^(z>>30))*0xBF58476D1CE4E5B9ull; z=(z^(z>>27))*0x94D049BB133111EBull; return z^(z>>31);}
int main(){
  using C=Ed25519;
  const Aff<C> g=aff_generator<C>();
  Aff<C> pts[16];
  for(int i=0;i<16;i++){ uint32_t k[8]; for(int t=0;t<8;t++)k[t]=(uint32_t)rnd(); k[7]&=0x0fffffff; Aff<C> a=jac_to_aff(aff_mul_words(g,k,8)); fe_cond_sub_p(a.x); fe_cond_sub_p(a.y); pts[i]=a; }
  Xyzz<C> lazy=xyzz_inf<C>(); Jac<C> eager=jac_inf<C>();
  for(int s=0;s<3000;s++){
    int i=rnd()%16; bool neg=rnd()&1;
    Aff<C> q=pts[i]; if(s%97==0) q=aff_inf<C>();
    xyzz_madd_lazy(lazy,q,neg);
    eager=jac_madd(eager, neg?aff_neg(q):q);
    if(!jac_eq(xyzz_to_jac(lazy),eager)){ printf("mismatch at %d\n",s); return 1; }
    // bounds: coordinates < 1.01p i.e. below 2p
    if(!(fe_below_kp<2>(lazy.e.X)&&fe_below_kp<2>(lazy.e.Y)&&fe_below_kp<2>(lazy.e.Z)&&fe_below_kp<2>(lazy.e.T))){printf("bound at %d\n",s);return 1;}
  }
  printf("ok ed25519 3000\n"); return 0;
}
