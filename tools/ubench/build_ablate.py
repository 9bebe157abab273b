import concurrent.futures, __graft_entry__ as g
def b(i): return g.build_variant('abl%d' % i, ['-DSKF_ABLATE=%d' % i])
with concurrent.futures.ThreadPoolExecutor(4) as ex: print(list(ex.map(b, [1,2,3,4])))
