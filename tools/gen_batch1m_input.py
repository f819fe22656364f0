import sys
sys.path.insert(0,'.')
import __graft_entry__ as ge
z=ge.load()
mix=("xorshift","itext","lowent4k")
with open('/tmp/batch1m.bin','wb') as f:
    for i in range(128):
        f.write(z.gen(mix[i%3],12345+i,1<<20).tobytes())
print("written")
