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


def _project(per_day, start_day, collection_day, obs_days):
    """per_day[day] = population per row item on oviposition day `day`; returns the
    expected emergences per item and observation date (Bayes_funcs.py:57-90, :116-144)."""
    nitem = len(per_day[start_day]) if collection_day > start_day else 0
    emerg_proj = np.zeros((nitem, max_incubation_time))
    for day in range(start_day, collection_day):
        max_post_col = day + max_incubation_time - collection_day
        min_post_col = max(0, max_post_col + 1 - incubation_time.size)
        span_len = max_post_col - min_post_col + 1
        e = np.outer(per_day[day], incubation_time)
        emerg_proj[:, min_post_col:max_post_col + 1] += e[:, -span_len:]
    col = obs_days - collection_day
    out = np.zeros((nitem, len(obs_days)))
    out[:, 0] = emerg_proj[:, 0:col[0] + 1].sum(axis=1)
    for n, c in enumerate(col[1:]):
        out[:, n + 1] = emerg_proj[:, col[n] + 1:c + 1].sum(axis=1)
    return out


def _gather_days(modelsol, days, rows, cols):
    '''{day: values at the cells}; one device call for all days when the model offers it.'''
    days = list(days)
    if not days:
        return {}
    if hasattr(modelsol, 'gather_days'):
        vals = modelsol.gather_days(days, rows, cols)
        return {day: vals[n] for n, day in enumerate(days)}
    return {day: modelsol.gather(day, rows, cols) for day in days}


def popdensity_to_emergence(modelsol, locinfo):
    '''Expected number of wasps per release-field grid point / sentinel field whose
    oviposition results in emergence on each observation date (Bayes_funcs.py:20-152).
    Returns (release_emerg, sentinel_emerg): one array per collection.'''
    release_emerg = []
    for nframe, dframe in enumerate(locinfo.release_DataFrames):
        collection_day = _days(locinfo.collection_datesPR[nframe])
        start_day = max(collection_day - max_incubation_time, 0)
        cells = np.asarray(locinfo.emerg_grids[nframe]).reshape(-1, 2)
        per_day = _gather_days(modelsol, range(start_day, collection_day), cells[:, 0], cells[:, 1])
        release_emerg.append(_project(per_day, start_day, collection_day, _unique_days(dframe)))
    sentinel_emerg = []
    for nframe, dframe in enumerate(locinfo.sent_DataFrames):
        collection_day = _days(locinfo.collection_datesPR[nframe])
        start_day = max(collection_day - max_incubation_time, 0)
        fields = [np.asarray(locinfo.field_cells[f]) for f in locinfo.sent_ids]
        rows = np.concatenate([f[:, 0] for f in fields])
        cols = np.concatenate([f[:, 1] for f in fields])
        bounds = np.cumsum([0] + [len(f) for f in fields])
        per_day = {day: np.array([v[bounds[i]:bounds[i + 1]].sum() for i in range(len(fields))])
                   for day, v in _gather_days(modelsol, range(start_day, collection_day), rows, cols).items()}
        sentinel_emerg.append(_project(per_day, start_day, collection_day, _unique_days(dframe)))
    return (release_emerg, sentinel_emerg)


def popdensity_grid(modelsol, locinfo):
    '''Expected number of wasps in each grid point on each grid observation date
    (Bayes_funcs.py:156-179); the model holds end-of-day results.'''
    cells = np.asarray(locinfo.grid_cells)
    out = np.zeros((cells.shape[0], len(locinfo.grid_obs_datesPR)))
    days = [_days(date) - 1 for date in locinfo.grid_obs_datesPR]
    vals = _gather_days(modelsol, sorted(set(days)), cells[:, 0], cells[:, 1])
    for nday, day in enumerate(days):
        out[:, nday] = vals[day]
    return out


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
