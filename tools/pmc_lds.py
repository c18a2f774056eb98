import csv,glob,sys,collections
d=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,v in agg.items():
    if 'pw_gemm' in k or 'conv3x3_fwd_w' in k:
        print(k, {c: round(x/ max(cnt[(k,c)],1)) for c,x in v.items()})
