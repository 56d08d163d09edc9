#! /usr/bin/env python3
'''Running parasitoid model simulations on the MI355X -- counterpart of the reference's
`Run.py`: the `Params` configuration object (same attributes, presets, `config.txt` /
command line / JSON handling, `get_model_params()` / `get_wind_params()` tuple orders,
Run.py:34-384) and `main(params)` (Run.py:388-520) with the day kernels built in one
device batch and the day chain run by `CalcSol` on the GPU.

Not carried over: plotting (`Plot_Result`, network/matplotlib) and the automatic creation of
a `config.txt` in the working directory.
'''
import json
import os
import sys
import time

import numpy as np
from scipy import sparse

from . import globalvars
from . import ParasitoidModel as PM
from .CalcSol import get_solutions, get_populations


def _tuple_of(types):
    def parse(val):
        parts = val.strip(' ()').split(',')
        return tuple(t(p) for t, p in zip(types, parts))
    return parse


def _flag(val):
    return {'True': True, 'False': False}.get(val, bool(val))


# dataset presets (Run.py:96-138): site_name, start_time, coord, r_dur, r_dist, r_start, r_number
_DATASETS = {
    None: ('data/carnarvonearl', '00:30', None, None, None, None, None),
    'carnarvon': ('data/carnarvonearl', '00:30', (-24.851614, 113.731267), 5, 'uniform', 0.354, 40000),
    'kalbar': ('data/kalbar', '00:00', (-27.947131, 152.584171), 1, 'uniform', None, 130000),
}

# key=value parameters (Run.py:263-352): attribute -> parser
_PARSERS = {
    'outfile': str, 'site_name': str, 'start_time': str, 'maps_key': str, 'maps_service': str,
    'coord': _tuple_of((float, float)),
    'domain_info': _tuple_of((float, int)),
    'interp_num': int, 'ndays': int, 'r_dur': int, 'n_periods': int, 'min_ndays': int,
    'g_params': _tuple_of((float, float)),
    'f_params': _tuple_of((float,) * 4),
    'Dparams': _tuple_of((float,) * 3),
    'Dlparams': _tuple_of((float,) * 3),
    'lam': float, 'mu_r': float,
}
_FLAG_KEYS = {'output': 'OUTPUT', 'plot': 'PLOT', 'cuda': 'CUDA'}
_OPTIONS = {
    'no_output': ('OUTPUT', False), 'output': ('OUTPUT', True),
    'no_plot': ('PLOT', False), 'plot': ('PLOT', True),
    'no_cuda': ('CUDA', False), 'cuda': ('CUDA', True),
}


class Params():
    '''Parameters of a model run (reference Run.py:34-384).'''
    ### Simulation flags ### (shared among all Params instances)
    OUTPUT = True
    PLOT = False     # plotting is not part of this package
    CUDA = True      # selects the device backend (name kept from the reference)

    def __init__(self, config='config.txt'):
        self.PROB_MODEL = True
        self.dataset = 'kalbar'
        self.my_datasets()
        # (dist (m), cells) from release point to side of domain
        self.domain_info = (10000.0, 400)
        self.interp_num = 30
        self.ndays = -1
        self.g_params = (1.263, 3.913)
        self.f_params = (7.302, 2.614, 23.999, 2.350)
        self.Dparams = (171.82, 144.58, 0.253)
        self.Dlparams = (7.096, 7.260, 0.000)
        self.lam = 1.
        self.mu_r = 1.179
        self.n_periods = 30
        self.maps_key = None
        self.maps_service = 'Google'
        self.min_ndays = 6
        self.default_chg(config)

    def my_datasets(self):
        if self.dataset not in _DATASETS:
            print('Unknown dataset in Params.dataset.')
        else:
            (self.site_name, self.start_time, self.coord, self.r_dur, self.r_dist,
             self.r_start, self.r_number) = _DATASETS[self.dataset]
        stem = 'output/' + (self.dataset if self.dataset is not None else '')
        if not self.PROB_MODEL:
            stem += '_pop' if self.dataset is not None else 'poprun'
        self.outfile = stem + time.strftime('%m%d-%H%M')

    ########    Methods for multiple-day emergence    ########

    def uniform(self, day):
        '''Uniform distribution over emergence days. 1 <= day <= self.r_dur.'''
        return 1. / self.r_dur

    def custom(self, day):
        pass

    def r_mthd(self):
        '''Function handle of the emergence distribution named by r_dist.'''
        return {'uniform': self.uniform, 'custom': self.custom}.get(self.r_dist)

    ########    Methods for changing parameters    ########

    def default_chg(self, config='config.txt'):
        '''Apply `key = value` lines of config.txt if it exists ("#" starts a comment).'''
        if not config or not os.path.exists(config):
            return
        try:
            with open(config) as f:
                for line in f:
                    line = line.split('#', 1)[0]
                    words = line.split('=')
                    if len(words) > 1:
                        self.chg_param(words[0].strip(), words[1].strip())
            self.my_datasets()
        except ValueError:
            print(' in config.txt.')
            raise

    def cmd_line_chg(self, args):
        '''Change parameters from command line args: --flag or <param name>=<new value>'''
        for argstr in args:
            if argstr.startswith('--'):
                opt = argstr[2:].lower()
                if opt in _OPTIONS:
                    setattr(self, *_OPTIONS[opt])
                elif opt in ('pop', 'popmodel', 'pop_model'):
                    self.PROB_MODEL = False
                    self.my_datasets()
                elif opt in ('prob', 'probmodel', 'prob_model'):
                    self.PROB_MODEL = True
                    self.my_datasets()
                elif opt in ('carnarvon', 'kalbar'):
                    self.dataset = opt
                    self.my_datasets()
                else:
                    raise ValueError('Unrecognized option {0}.'.format(argstr))
            else:
                arg, _eq, val = argstr.partition('=')
                self.chg_param(arg, val)

    def chg_param(self, arg, val):
        '''Change the parameter arg to val, where both are given as strings'''
        try:
            if arg.lower() == 'prob_model':
                self.prob_model = bool(val)       # (sic) reference Run.py:267-269
                self.my_datasets()
            elif arg == 'dataset':
                self.dataset = val
                self.my_datasets()
            elif arg in ('r_start', 'r_number'):
                # the reference compares instead of assigning (Run.py:292-295): the value
                # is parsed (errors surface) and the preset is kept
                (float if arg == 'r_start' else int)(val)
            elif arg in _PARSERS:
                setattr(self, arg, _PARSERS[arg](val))
            elif arg in _FLAG_KEYS:
                setattr(self, _FLAG_KEYS[arg], _flag(val))
            else:
                raise LookupError('Unrecognized parameter {0}.'.format(arg))
        except LookupError:
            print('Could not parse {0}.'.format(arg) + '\n ')
            raise
        except ValueError:
            print('Could not parse {0}.'.format(arg) +
                  ' Try enclosing this argument in quotations.\n ')
            raise

    def file_read_chg(self, filename):
        '''Read in parameters from a json file written by main()'''
        if filename.rstrip()[-5:] != '.json':
            filename = filename.rstrip() + '.json'
        try:
            with open(filename) as fobj:
                param_dict = json.load(fobj)
        except FileNotFoundError:
            print('Could not open file {0}.'.format(filename))
            raise
        for key in param_dict:
            setattr(self, key, param_dict[key])

    ########    Methods for getting function parameters    ########

    def get_model_params(self):
        '''Params in the order of the prob_mass signature, minus day & wind_data'''
        hparams = (self.lam, *self.g_params, *self.f_params)
        return (hparams, self.Dparams, self.Dlparams, self.mu_r, self.n_periods,
                *self.domain_info)

    def get_wind_params(self):
        '''Wind params to pass to PM.get_wind_data'''
        return (self.site_name, self.interp_num, self.start_time)


def recentre(pmf, rad_res):
    '''Shift a shrunk day kernel back into the dom_len x dom_len domain (Run.py:454-458).'''
    pmf = sparse.coo_matrix(pmf)
    offset = rad_res - pmf.shape[0] // 2
    dom_len = rad_res * 2 + 1
    return sparse.coo_matrix((pmf.data, (pmf.row + offset, pmf.col + offset)),
                             shape=(dom_len, dom_len))


def run_model(params, verbose=True):
    '''The compute part of main(): wind -> day kernels (one device batch) -> day chain.
    Returns (modelsol, days, ndays, timings).'''
    globalvars.cuda = bool(params.CUDA)
    wind_data, days = PM.get_wind_data(*params.get_wind_params())
    ndays = min(params.ndays, len(days)) if params.ndays >= 0 else len(days)
    say = print if verbose else (lambda *a, **k: None)

    tic = time.time()
    say("Calculating each day's spread on the GPU...")
    starts = [None] * ndays
    if not params.PROB_MODEL and ndays > 0:
        starts[0] = params.r_start
    pmf_list = PM.prob_mass_batch(days[:ndays], wind_data, *params.get_model_params(),
                                  start_times=starts)
    max_shape = np.array([0, 0])
    for pmf in pmf_list:
        max_shape = np.maximum(max_shape, pmf.shape)
    t_pm = time.time() - tic
    say('Time elapsed: {0}'.format(t_pm))

    rad_res = params.domain_info[1]
    dom_len = rad_res * 2 + 1
    tic = time.time()
    if params.PROB_MODEL:
        modelsol = [recentre(pmf_list[0], rad_res)]
        get_solutions(modelsol, pmf_list, days, ndays, dom_len, max_shape)
    else:
        r_spread = [recentre(pmf_list[ii], rad_res).tocsr() for ii in range(params.r_dur)]
        modelsol = get_populations(r_spread, pmf_list, days, ndays, dom_len, max_shape,
                                   params.r_dur, params.r_number, params.r_mthd())
    t_sol = time.time() - tic
    say('Done.')
    say('Time elapsed: {0}'.format(t_sol))
    return modelsol, days, ndays, {'prob_mass_s': t_pm, 'solver_s': t_sol}


def save_result(params, modelsol, days, ndays):
    '''Per-day CSR triplets + `days` in an .npz and the parameters as .json, the format the
    reference writes (Run.py:490-516) and its Plot_Result reads (Plot_Result.py:511-524).'''
    out = {}
    for n, day in enumerate(days[:ndays]):
        sol = modelsol[n].tocsr()
        out[str(day) + '_data'] = sol.data
        out[str(day) + '_ind'] = sol.indices
        out[str(day) + '_indptr'] = sol.indptr
    out['days'] = days[:ndays]
    dir_file = params.outfile.rsplit('/', 1)
    if len(dir_file) > 1 and not os.path.exists(dir_file[0]):
        os.makedirs(dir_file[0])
    np.savez(params.outfile, **out)
    with open(params.outfile + '.json', 'w') as fobj:
        param_dict = dict(params.__dict__)
        param_dict.pop('maps_key', None)
        json.dump(param_dict, fobj)


def main(params):
    '''Main routine for running model simulations; requires a Params object.'''
    tic_total = time.time()
    modelsol, days, ndays, _t = run_model(params)
    print('Total time elapsed: {0}'.format(time.time() - tic_total))
    if params.OUTPUT:
        print('Saving...')
        save_result(params, modelsol, days, ndays)
    if params.PLOT:
        print('Plotting is not part of parasitoids_amd; load {0}.npz with the reference '
              'Plot_Result.'.format(params.outfile))
    return modelsol


if __name__ == "__main__":
    params = Params()
    if len(sys.argv[1:]) > 0:
        params.cmd_line_chg(sys.argv[1:])
    main(params)
