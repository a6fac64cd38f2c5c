import csv,glob,sys,collections
for d in sys.argv[1:]:
    f=glob.glob(d+"/**/*counter_collection.csv",recursive=True)
    if not f: print(d,"no csv"); continue
    acc=collections.defaultdict(lambda: collections.Counter())
    for r in csv.DictReader(open(f[0])):
        n=r["Kernel_Name"]
        if "bf3" not in n or "split" in n: continue
        key="bf3p" if "bf3p" in n else "bf3"
        acc[key][r["Counter_Name"]]+=float(r["Counter_Value"]); acc[key]["_n_"+r["Counter_Name"]]+=1
    for k,c in acc.items():
        print(d.split("/")[-1],k,{a:("%.4g"%b) for a,b in c.items() if not a.startswith("_n_")})
