import os, sys, time, warnings
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
warnings.simplefilter('ignore')
from parasitoids_amd import Run, globalvars
R = int(sys.argv[1]); mode = sys.argv[2]
globalvars.fft_mode = mode
p = Run.Params(config=None)
p.cmd_line_chg(['--carnarvon', '--prob', 'domain_info=(10000.0,%d)' % R])
p.site_name = os.path.join('parasitoids_amd', p.site_name); p.OUTPUT = False
for rep in range(2):
    t0 = time.time()
    modelsol, days, ndays, t = Run.run_model(p, verbose=False)
    print('R=%d mode=%s rep=%d ndays=%d total %.2fs prob_mass %.2fs solver(incl. COO fetch) %.2fs' % (R, mode, rep, ndays, time.time()-t0, t['prob_mass_s'], t['solver_s']))
print('sums', [round(float(s.sum()), 12) for s in modelsol[:3]], '...', round(float(modelsol[-1].sum()), 12), 'nnz last', modelsol[-1].nnz)
