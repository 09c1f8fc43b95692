#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static uint64_t nxt(void){ uint64_t a=s[0], b=s[1]; s[0]=b; a^=a<<23; s[1]=a^b^(a>>17)^(b>>26); return s[1]+b; }
static double mk(uint64_t mant, int e){ uint64_t u = ((uint64_t)(e+1023)<<52) | (mant & 0xFFFFFFFFFFFFFull); double d; memcpy(&d,&u,8); return d; }
static inline double fast(double a, double b, double y){ double q0=a*y; double r=fma(-q0,b,a); return fma(r,y,q0); }
int main(int argc,char**argv){
    long n = atol(argv[1]); long bad=0;
    for(long i=0;i<n;i++){
        uint64_t ma=nxt(), mb=nxt(); int mode = i & 7;
        if(mode==1) mb = 0xFFFFFFFFFFFFFull;            /* all ones */
        if(mode==2) mb = 0xFFFFFFFFFFFFFull - (nxt()&0xff);
        if(mode==3) mb = nxt()&0xff;                     /* near power of two */
        if(mode==4) ma = 0xFFFFFFFFFFFFFull - (nxt()&0xff);
        if(mode==5) ma = nxt()&0xff;
        if(mode==6) { ma = nxt() & 0xFFFFFFF000000ull; mb = nxt() & 0xFFFFFF0000000ull; }   /* short mantissas: exact/halfway cases */
        double b = mk(mb, -(int)(nxt()%60));   /* |v| in (2^-60, 2) */
        if (b > 1.0) b *= 0.5;
        double a = mk(ma, (int)(nxt()%140)-70);
        if(nxt()&1) a=-a; if(nxt()&1) b=-b;
        double y = 1.0/b;
        double q = a/b, f = fast(a,b,y);
        if(q!=f){ if(bad<10) printf("MISMATCH a=%a b=%a q=%a f=%a\n",a,b,q,f); bad++; }
    }
    printf("n=%ld mismatches=%ld\n", n, bad);
    return 0;
}
