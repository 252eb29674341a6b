import csv,glob,collections,sys
d=sys.argv[1]
f=sorted(glob.glob(d+"/**/*counter_collection.csv",recursive=True))[-1]
agg=collections.defaultdict(lambda:collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].replace("(anonymous namespace)::","")[:44]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
names=sorted({c for v in agg.values() for c in v})
print("kernel".ljust(46)+" ".join(n[-16:].rjust(16) for n in names))
for k,v in sorted(agg.items(),key=lambda x:-sum(x[1].values()))[:12]:
    print(k.ljust(46)+" ".join(("%.3e"%v[n]).rjust(16) for n in names))
