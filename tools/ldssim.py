# LDS bank-conflict model of cab_phase1r_kernel's access patterns (MI355X_MICROARCH.md LDS table)
import itertools, sys
def groups_read_b128():
    base=[list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32))]
    return base+[[l+32 for l in gset] for gset in base]
def groups_contig(n): return [list(range(i,i+n)) for i in range(0,64,n)]
def cycles(addrs, width, groups, nbanks, active=None):
    tot=0
    for gset in groups:
        bank={}
        for l in gset:
            if active is not None and not active[l]: continue
            a=addrs[l]
            for d in range(width//4):
                b=((a//4)+d)%nbanks
                bank.setdefault(b,set()).add((a+4*d)//4)
        tot+=max([len(v) for v in bank.values()] or [0])
    return tot
def rd128(addrs,active=None): return cycles(addrs,16,groups_read_b128(),64,active)
def wr64(addrs,active=None): return cycles(addrs,8,groups_contig(16),32,active)
def wr128(addrs,active=None): return cycles(addrs,16,groups_contig(8),32,active)

def shape(C,HW,NSW=None):
    NGP=C//16; NSW=NSW or (4 if NGP==4 else 2)
    CH=C//2; K=C+CH if HW else C; KS1=(K+2+31)//32; KS2=(C+31)//32
    PSX=KS1*64+32; XPL=16*PSX+32; XSLOT=4*XPL; GPL=18*16; GROW=NGP*2*4*GPL; PSR=160; RSLOT=64*PSR+64; PSO=C*2+16; OSLOT=64*PSO
    return dict(NGP=NGP,NSW=NSW,CH=CH,K=K,KS1=KS1,KS2=KS2,PSX=PSX,XPL=XPL,XSLOT=XSLOT,GPL=GPL,GROW=GROW,PSR=PSR,RSLOT=RSLOT,PSO=PSO,OSLOT=OSLOT)

def report(C,HW,over=None):
    S=shape(C,HW)
    if over: S.update(over)
    g_=lambda l:l>>4; p_=lambda l:l&15
    out={}
    # A: x reads
    tot=0;n_=0
    for n in range(4):
        for s in range(S['KS1']):
            a=[p_(l)*S['PSX']+g_(l)*16+n*S['XPL']+64*s for l in range(64)]
            tot+=rd128(a); n_+=1
    out['A x read b128']=(n_,tot,n_*4)
    # A: g1 writes
    tot=0;n_=0
    for q in range(1):
        for n in range(4):
            a=[((2*q+(g_(l)>>1))*4)*S['GPL']+(p_(l)+1)*16+(g_(l)&1)*8+n*S['GPL'] for l in range(64)]
            tot+=wr64(a); n_+=1
    out['A g1 write b64']=(n_,tot,n_*4)
    # B: gemm2 reads
    tot=0;n_=0
    for n in range(4):
        for s in range(S['KS2']):
            a=[p_(l)*S['PSR']+g_(l)*16+n*16*S['PSR']+64*s for l in range(64)]
            tot+=rd128(a); n_+=1
    out['B r read b128']=(n_,tot,n_*4)
    # B: repconv reads, averaged over jm
    tot=0;n_=0
    for jm in range(6):
        for i in range(32):
            s_=i>>2;G=(i>>1)&1;uu=i&1
            a=[]
            for l in range(64):
                g=g_(l);p=p_(l); gq0=0*2*4*S['GPL']+p*16
                if s_<6:
                    r6=(jm+g)%6; e=2*uu+s_-2+4
                    a.append(r6*S['GROW']+gq0+(G*4+(e&3))*S['GPL']+(e>>2)*16)
                else:
                    s7=s_-6; dx6=4+(g&1) if s7 else g; e=2*uu+dx6-2+4
                    r4=(jm+4)%6
                    a.append(r4*S['GROW']+gq0+(e&3)*S['GPL']+(e>>2)*16+G*4*S['GPL'])
            tot+=rd128(a); n_+=1
    out['B g1 read b128 (repconv)']=(n_/6,tot/6,n_/6*4)
    # B: r writes
    tot=0;n_=0
    for uu in range(2):
        for G in range(2):
            a=[p_(l)*S['PSR']+(16*0+4*(g_(l)&1))*2+(2*uu+(g_(l)>>1))*16*S['PSR']+16*G for l in range(64)]
            tot+=wr64(a); n_+=1
    out['B r write b64']=(n_,tot,n_*4)
    # B: out writes
    tot=0;n_=0
    for n in range(4):
        a=[p_(l)*S['PSO']+(16*0+4*g_(l))*2+n*16*S['PSO'] for l in range(64)]
        tot+=wr64(a); n_+=1
    out['B out write b64']=(n_,tot,n_*4)
    # S: x writes
    LPP=S['NSW']; NPC=S['K']//8; NP0=(NPC+LPP-1)//LPP
    tot=0;n_=0;cnt=0
    for q in range(S['NSW']):
        for i in range(NP0):
            a=[];act=[]
            for l in range(64):
                stid=q*64+l; quad=stid//LPP; sub=stid%LPP
                spx=((quad&~3)|((quad&1)<<1)|((quad>>1)&1)) if LPP==4 else quad
                xpix=(spx&3)*S['XPL']+(spx>>2)*S['PSX']
                a.append(xpix+sub*16+i*16*LPP); act.append(LPP*i+sub<NPC)
            tot+=wr128(a,act); n_+=1
    out['S x write b128 (all stagers)']=(n_,tot,n_*8)
    # S: out reads
    NSTH=64*S['NSW']; NPO=C//8; NIT=(61*NPO+NSTH-1)//NSTH
    tot=0;n_=0
    for q in range(S['NSW']):
        for k in range(NIT):
            a=[];act=[]
            for l in range(64):
                e=q*64+l+NSTH*k; px=e//NPO; pc=e-px*NPO; rc=3+px
                ok=rc<61
                a.append(((rc&3)*16+(rc>>2))*S['PSO']+pc*16 if ok else 0); act.append(True)
            tot+=rd128(a,act); n_+=1
    out['S out read b128 (all stagers)']=(n_,tot,n_*4)
    return S,out
if __name__=='__main__':
    for C,HW in ((64,True),(64,False),(80,True),(80,False)):
        S,o=report(C,HW)
        print(f"C={C} HW={HW} PSX={S['PSX']} XPL={S['XPL']} GROW={S['GROW']} PSO={S['PSO']}")
        NGP=S['NGP']
        tot_c=tot_b=0
        for k,(n,c,b) in o.items():
            mult = 1 if 'all stagers' in k else NGP
            print(f"  {k:34s} n={n:5.1f} cycles={c:7.1f} ideal={b:6.1f}  x{mult} waves -> extra {mult*(c-b):7.1f}")
            tot_c+=mult*c; tot_b+=mult*b
        print(f"  per step: {tot_c:.0f} cycles, ideal {tot_b:.0f}, conflicts {tot_c-tot_b:.0f} ({(tot_c-tot_b)/tot_c:.1%})")
