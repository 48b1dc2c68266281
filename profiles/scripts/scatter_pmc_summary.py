"""FETCH_SIZE / WRITE_SIZE of the standalone aggregation kernels (gpurun_out/r02sc, profiles/scripts/pmc_scatter.sh) against the
algorithmic bytes of profiles/r02_scatter_bench.json -> profiles/r02_scatter_pmc_hbm.csv"""
import csv,glob,collections,json
def load(d):
    f=glob.glob(f'/root/repo/gpurun_out/r02sc/{d}/**/*counter_collection.csv',recursive=True)[0]
    per=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'gin_gather' in r['Kernel_Name'] or 'gine_gather' in r['Kernel_Name']:
            per[(r['Kernel_Name'].split('(')[0], int(r['Grid_Size']))].append(float(r['Counter_Value']))
    return per
fe=load('pmc_fetch'); wr=load('pmc_write')
d=json.load(open('/root/repo/profiles/r02_scatter_bench.json'))
alg={}
for k,v in d['kernels'].items():
    alg[('gine' if 'gine' in k else 'gin', v['graphs'])]=v['algorithmic_bytes']
graphs=[128,1024,2048,6144]
with open('/root/repo/profiles/r02_scatter_pmc_hbm.csv','w') as out:
    out.write("kernel,graphs,grid,dispatches,FETCH_SIZE_KiB_mean,WRITE_SIZE_KiB_mean,hbm_bytes_2xFETCH_plus_WRITE,algorithmic_bytes,ratio\n")
    for kind in ('gin_gather','gine_gather'):
        ks=sorted([k for k in fe if kind+'<' in k[0]], key=lambda k:k[1])
        for g,k in zip(graphs,ks):
            f=sum(fe[k])/len(fe[k]); w=sum(wr[k])/len(wr[k]); b=(2*f+w)*1024; a=alg[('gine' if 'gine' in kind else 'gin', g)]
            line='"%s",%d,%d,%d,%.0f,%.0f,%d,%d,%.3f'%(k[0],g,k[1],len(fe[k]),f,w,b,a,b/a); out.write(line+"\n"); print(line[-60:])
for k,v in d['kernels'].items(): print(k, round(v['mean_launch_us'],1), round(v['achieved']), round(v['frac'],3))
