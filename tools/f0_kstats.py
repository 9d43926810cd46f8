import csv,glob,sys
for d in sys.argv[1:]:
    f=glob.glob(d+'/*/*kernel_stats.csv')[0]
    out=[]
    for r in csv.DictReader(open(f)):
        if 'f0' in r['Name'] or 'logs' in r['Name']:
            out.append('%s %.2f'%(r['Name'].split('(')[0].replace('afx::','').replace('void ','')[:14], float(r['AverageNs'])/1e6))
    print(d.split('/')[-1], ' | '.join(out))
