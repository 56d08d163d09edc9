"""Population solution -> expected observations, counterpart of the reference's
`Bayes_funcs.py` (same function names; `locinfo` is the reference's `Data_Import.LocInfo`
or any object with the attributes read below).

`modelsol` is a device-resident evaluation (`pop_model.PopModel` after `evaluate`, or any
object with `.gather(day, rows, cols)`): the daily fields stay on the GPU and only the
cells an observation needs are gathered (`ps_record_gather`), instead of building 18 CSR
matrices on the host and indexing them one cell at a time (Bayes_funcs.py:65-71, :122-128,
:170-172).  The projection arithmetic on the gathered values is a few hundred numbers and
stays in numpy.
"""
import numpy as np

### Oviposition to emergence time (Bayes_funcs.py:10-18): incubation is 19 to 25 days
incubation_time = np.array([0.05, 0.1, 0.2, 0.3, 0.2, 0.1, 0.05])
max_incubation_time = 25


def _days(t):
    return int(getattr(t, 'days', t))


def _unique_days(dframe):
    seen, out = set(), []
    for t in dframe['datePR']:
        d = _days(t)
        if d not in seen:
            seen.add(d)
            out.append(d)
    return np.array(out)


def _projection_matrix(start_day, collection_day, obs_days):
    """W[day - start_day, n]: share of the oviposition of `day` that emerges in the n-th
    observation interval after the collection (Bayes_funcs.py:57-90, :116-144) -- the
    incubation-time spread followed by the binning into observation dates, as one matrix, so
    that expected emergences = per_day.T @ W."""
    nday = max(collection_day - start_day, 0)
    emerg = np.zeros((nday, max_incubation_time))
    for day in range(start_day, collection_day):
        max_post_col = day + max_incubation_time - collection_day
        min_post_col = max(0, max_post_col + 1 - incubation_time.size)
        span_len = max_post_col - min_post_col + 1
        emerg[day - start_day, min_post_col:max_post_col + 1] += incubation_time[-span_len:]
    col = np.asarray(obs_days) - collection_day
    W = np.zeros((nday, len(col)))
    W[:, 0] = emerg[:, 0:col[0] + 1].sum(axis=1)
    for n, c in enumerate(col[1:]):
        W[:, n + 1] = emerg[:, col[n] + 1:c + 1].sum(axis=1)
    return W


def _plan(locinfo):
    """Everything of popdensity_to_emergence / popdensity_grid that depends on the site data
    only -- collection and observation days, cell lists, field boundaries, projection
    matrices -- computed once per `locinfo` object (it is static during a sampling run; the
    per-evaluation work is then one device gather and a few small matrix products)."""
    plan = getattr(locinfo, '_ps_plan', None)
    if plan is not None:
        return plan
    rel = []
    for nframe, dframe in enumerate(locinfo.release_DataFrames):
        collection_day = _days(locinfo.collection_datesPR[nframe])
        start_day = max(collection_day - max_incubation_time, 0)
        cells = np.asarray(locinfo.emerg_grids[nframe]).reshape(-1, 2)
        rel.append(dict(days=list(range(start_day, collection_day)), rows=cells[:, 0], cols=cells[:, 1],
                        W=_projection_matrix(start_day, collection_day, _unique_days(dframe))))
    sen = []
    for nframe, dframe in enumerate(locinfo.sent_DataFrames):
        collection_day = _days(locinfo.collection_datesPR[nframe])
        start_day = max(collection_day - max_incubation_time, 0)
        fields = [np.asarray(locinfo.field_cells[f]) for f in locinfo.sent_ids]
        sen.append(dict(days=list(range(start_day, collection_day)),
                        rows=np.concatenate([f[:, 0] for f in fields]),
                        cols=np.concatenate([f[:, 1] for f in fields]),
                        starts=np.cumsum([0] + [len(f) for f in fields])[:-1],
                        empty=np.array([len(f) == 0 for f in fields]),
                        W=_projection_matrix(start_day, collection_day, _unique_days(dframe))))
    cells = np.asarray(locinfo.grid_cells)
    gdays = [_days(date) - 1 for date in locinfo.grid_obs_datesPR]
    grid = dict(rows=cells[:, 0], cols=cells[:, 1], days=gdays, udays=sorted(set(gdays)))
    plan = dict(rel=rel, sen=sen, grid=grid)
    # one gather for everything: the union of the days, all cell lists back to back
    parts = rel + sen + [dict(days=grid['udays'], rows=grid['rows'], cols=grid['cols'])]
    udays = sorted(set(d for p in parts for d in p['days']))
    pos = {d: n for n, d in enumerate(udays)}
    off = np.cumsum([0] + [len(p['rows']) for p in parts])
    plan['all'] = dict(days=udays, rows=np.concatenate([np.asarray(p['rows']) for p in parts]).astype(np.int32),
                       cols=np.concatenate([np.asarray(p['cols']) for p in parts]).astype(np.int32),
                       slices=[(np.array([pos[d] for d in p['days']], dtype=int), int(off[i]), int(off[i + 1]))
                               for i, p in enumerate(parts)])
    try:
        locinfo._ps_plan = plan
    except AttributeError:
        pass
    return plan


def _gather_matrix(modelsol, days, rows, cols):
    """[len(days), len(rows)] values at the cells; one device call when the model offers it."""
    days = list(days)
    if not days:
        return np.zeros((0, len(rows)))
    if hasattr(modelsol, 'gather_days'):
        return np.asarray(modelsol.gather_days(days, rows, cols))
    return np.array([modelsol.gather(day, rows, cols) for day in days])


def _gather_days(modelsol, days, rows, cols):
    '''{day: values at the cells}; one device call for all days when the model offers it.'''
    days = list(days)
    return dict(zip(days, _gather_matrix(modelsol, days, rows, cols)))


def _field_sums(vals, fr):
    """[day, field] sums of the [day, field cell] values over each sentinel field's cells
    (Bayes_funcs.py:116-144 sums the slice of every field).  A field polygon may hold no cell
    centre at coarse rad_res: reduceat runs on the starts of the NON-empty fields only -- those are
    strictly increasing and inside the array, and the cells between two of them are exactly one
    field's (an empty field owns none) -- and the empty fields stay zero."""
    out = np.zeros((vals.shape[0], len(fr['starts'])))
    live = ~np.asarray(fr['empty'], dtype=bool)
    if vals.shape[0] and vals.shape[1] and live.any():
        out[:, live] = np.add.reduceat(vals, np.asarray(fr['starts'])[live], axis=1)
    return out


def popdensity_to_emergence(modelsol, locinfo):
    '''Expected number of wasps per release-field grid point / sentinel field whose
    oviposition results in emergence on each observation date (Bayes_funcs.py:20-152).
    Returns (release_emerg, sentinel_emerg): one array per collection.'''
    plan = _plan(locinfo)
    release_emerg = []
    for fr in plan['rel']:
        per_day = _gather_matrix(modelsol, fr['days'], fr['rows'], fr['cols'])      # [day, grid point]
        release_emerg.append(per_day.T @ fr['W'])
    sentinel_emerg = []
    for fr in plan['sen']:
        vals = _gather_matrix(modelsol, fr['days'], fr['rows'], fr['cols'])         # [day, field cell]
        sentinel_emerg.append(_field_sums(vals, fr).T @ fr['W'])
    return (release_emerg, sentinel_emerg)


def expected_observations(modelsol, locinfo):
    """(release_emerg, sentinel_emerg, grid_counts) -- popdensity_to_emergence and popdensity_grid
    (Bayes_Run.py:325-336) from ONE device gather: every cell list of the site against the union
    of the days any of them needs, sliced on the host."""
    plan = _plan(locinfo)
    if not hasattr(modelsol, 'gather_days'):
        rel, sen = popdensity_to_emergence(modelsol, locinfo)
        return rel, sen, popdensity_grid(modelsol, locinfo)
    al = plan['all']
    vals = np.asarray(modelsol.gather_days(al['days'], al['rows'], al['cols']))
    blocks = [vals[np.ix_(didx, np.arange(lo, hi))] if len(didx) else np.zeros((0, hi - lo))
              for didx, lo, hi in al['slices']]
    nrel, nsen = len(plan['rel']), len(plan['sen'])
    release_emerg = [blocks[i].T @ plan['rel'][i]['W'] for i in range(nrel)]
    sentinel_emerg = []
    for i, fr in enumerate(plan['sen']):
        v = blocks[nrel + i]
        sentinel_emerg.append(_field_sums(v, fr).T @ fr['W'])
    g = plan['grid']
    index = {day: n for n, day in enumerate(g['udays'])}
    grid = blocks[-1][[index[day] for day in g['days']]].T.copy()
    return release_emerg, sentinel_emerg, grid


def popdensity_grid(modelsol, locinfo):
    '''Expected number of wasps in each grid point on each grid observation date
    (Bayes_funcs.py:156-179); the model holds end-of-day results.'''
    g = _plan(locinfo)['grid']
    vals = _gather_matrix(modelsol, g['udays'], g['rows'], g['cols'])
    index = {day: n for n, day in enumerate(g['udays'])}
    return vals[[index[day] for day in g['days']]].T.copy()


def popdensity_card(modelsol, locinfo, domain_info):
    '''Expected number of wasps along the cardinal directions (north, south, east, west rows)
    at the sampled distances (Bayes_funcs.py:183-221).'''
    res = domain_info[0] / domain_info[1]
    c0 = int(domain_info[1])
    card_counts = []
    for nday, date in enumerate(locinfo.card_obs_datesPR):
        obslen = locinfo.card_obs[nday].shape[1]
        dist = 5 + locinfo.step_size[nday] * np.arange(1, obslen + 1)
        delta = (dist // res).astype(int)
        rows = np.concatenate([c0 - delta, c0 + delta, np.full(obslen, c0), np.full(obslen, c0)])
        cols = np.concatenate([np.full(obslen, c0), np.full(obslen, c0), c0 + delta, c0 - delta])
        card_counts.append(modelsol.gather(_days(date) - 1, rows, cols).reshape(4, obslen))
    return card_counts
