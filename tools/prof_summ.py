import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0][-60:]
    print(f"{int(r['Calls']):5d} {float(r['AverageNs'])/1e3:10.1f} us {float(r['Percentage']):6.2f}% {n}")
