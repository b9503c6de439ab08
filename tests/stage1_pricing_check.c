/*
 *  stage1_pricing_check.c -- test helper (CPU): the restructured stage-1 position pricing of the
 *  device scan (fiasco_amd/csrc/hip/mp_device.inc: StepCtx, mp_step_prepare, stage1) against
 *  the straightforward rle_bits walk over the merged position list (reference
 *  codec/domain-pool.c:737-793 as called from codec/approx.c:433-458), bit for bit, on random
 *  models: N positions, up to 4 kept vectors, optional luminance-state position, every
 *  candidate class (position 0, the luminance state, N-1, N-2, all intervals).
 *  usage: stage1_pricing_check [iterations]      exit status 0 = no mismatch
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define MAXED 5
static unsigned bbc(unsigned value, unsigned maxval){unsigned k=31u-(unsigned)__builtin_clz(maxval+1);unsigned r=(maxval+1)-(1u<<k);return value<maxval+1-2*r?k:k+1;}
typedef struct { int np; int p[MAXED-1]; float L,Ly,Q0,Q1; int ypos; unsigned N;
  float pre[MAXED]; int last[MAXED],k[MAXED],thr[MAXED]; unsigned cd,has; float sfx[MAXED]; float z0,zy; } Ctx;
#define RLE_EMIT(x) do{unsigned x_=(unsigned)(x); if(x_&&(N-1-last)){bits+=(float)bbc(x_-last,N-1-last);last=x_+1;}}while(0)
static float old_bits(const Ctx*c,int d){
  const unsigned N=c->N; int isy=d==c->ypos; int uses0=(d==0&&!isy)||(c->np>0&&c->p[0]==0);
  float bits=isy?c->Ly:c->L; bits+=uses0?c->Q1:c->Q0; unsigned last=1; int placed=isy;
  for(int i=0;i<MAXED-1;i++) if(i<c->np){int pi=c->p[i]; if(!placed&&d<pi){RLE_EMIT(d);placed=1;} RLE_EMIT(pi);}
  if(!placed) RLE_EMIT(d);
  return bits;
}
static void prepare(Ctx*c){
  const unsigned N=c->N; const int np=c->np; const int*p=c->p;
  int p0zero=np>0&&p[0]==0;
  float e[MAXED]; int ev[MAXED];
  for(int j=0;j<np;j++){
    if(j==0){ ev[0]=p[0]!=0&&(N-2)!=0; e[0]=ev[0]?(float)bbc((unsigned)p[0]-1,N-2):0; }
    else { ev[j]=(unsigned)p[j-1]!=N-2; e[j]=ev[j]?(float)bbc((unsigned)(p[j]-p[j-1]-1),N-2-(unsigned)p[j-1]):0; }
  }
  float b=c->L; b+=p0zero?c->Q1:c->Q0;
  c->cd=0;c->has=0;
  for(int i=0;i<=np;i++){
    c->pre[i]=b;
    unsigned last=i==0?1u:(unsigned)p[i-1]+1;
    c->last[i]=(int)last;
    unsigned mv=N-1-last;
    if(mv!=0){ c->cd|=1u<<i; unsigned k=31u-(unsigned)__builtin_clz(mv+1); unsigned r=(mv+1)-(1u<<k); c->k[i]=(int)k; c->thr[i]=(int)(mv+1-2*r);} else {c->k[i]=0;c->thr[i]=0;}
    if(i<np&&ev[i]) b+=e[i];
  }
  for(int j=1;j<np;j++){ c->sfx[j]=e[j]; if(ev[j]) c->has|=1u<<j; }
  b=c->L; b+=c->Q1; for(int j=0;j<np;j++) if(ev[j]) b+=e[j]; c->z0=b;
  b=c->Ly; b+=p0zero?c->Q1:c->Q0; for(int j=0;j<np;j++) if(ev[j]) b+=e[j]; c->zy=b;
}
static float new_bits(const Ctx*c,int d){
  float bits=c->pre[0]; int last=c->last[0],k=c->k[0],thr=c->thr[0]; int cd=c->cd&1; int nextp=c->np>0?c->p[0]:-1; int i=0;
  for(int j=0;j<MAXED-1;j++) if(j<c->np&&d>c->p[j]){i=j+1;bits=c->pre[j+1];last=c->last[j+1];k=c->k[j+1];thr=c->thr[j+1];cd=(c->cd>>(j+1))&1;nextp=(j+1<c->np)?c->p[j+1]:-1;}
  if(cd) bits+=(float)((d-last<thr)?k:k+1);
  unsigned mv=c->N-2u-(unsigned)d;
  if(nextp>=0&&mv!=0){ bits+=(float)bbc((unsigned)(nextp-d-1),mv); for(int j=1;j<MAXED-1;j++) if(j>i&&j<c->np&&((c->has>>j)&1)) bits+=c->sfx[j]; }
  if(d==0) bits=c->z0;
  if(d==c->ypos) bits=c->zy;
  return bits;
}
int main(int argc,char**argv){
  long iters=argc>1?atol(argv[1]):3000000;
  srand(1); long n=0,bad=0;
  for(long it=0;it<iters;it++){
    Ctx c; memset(&c,0,sizeof c);
    c.N=2+rand()%((it%3==0)?6:(it%3==1?70:6000));
    int maxnp=c.N-1<4?c.N-1:4; c.np=rand()%(maxnp+1);
    c.ypos=(rand()%3==0)?rand()%c.N:-1;
    /* pick np distinct positions excluding ypos */
    int used[8]; int cnt=0; 
    while(cnt<c.np){int v=rand()%c.N; int ok=v!=c.ypos; for(int j=0;j<cnt;j++) if(used[j]==v) ok=0; if(ok) used[cnt++]=v; else if(c.N<=(unsigned)c.np+1&&rand()%50==0){c.np=cnt;break;}}
    for(int a=0;a<c.np;a++)for(int b2=a+1;b2<c.np;b2++) if(used[b2]<used[a]){int t=used[a];used[a]=used[b2];used[b2]=t;}
    for(int j=0;j<c.np;j++) c.p[j]=used[j];
    c.L=3.7f+rand()%100*0.013f; c.Ly=2.9f+rand()%100*0.017f; c.Q0=0.0113f*(rand()%50); c.Q1=1.0f+rand()%9;
    prepare(&c);
    for(int t=0;t<8;t++){ int d=rand()%c.N; if(t==0) d=0; if(t==1&&c.ypos>=0) d=c.ypos; if (t==2) d=c.N-1; if(t==3&&c.N>=2) d=c.N-2;
      int clash=0; for(int j=0;j<c.np;j++) if(c.p[j]==d) clash=1; if(clash) continue;
      float a=old_bits(&c,d), b=new_bits(&c,d); n++;
      if(memcmp(&a,&b,4)){ if(bad<10) printf("MISMATCH N=%u np=%d p=%d,%d,%d,%d ypos=%d d=%d old=%.9g new=%.9g\n",c.N,c.np,c.p[0],c.p[1],c.p[2],c.p[3],c.ypos,d,a,b); bad++; }
    }
  }
  printf("%ld cases, %ld mismatches\n",n,bad); return bad!=0;
}
