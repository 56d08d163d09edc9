"""Observation data of a release site -> the arrays the Bayesian observation model compares
with, counterpart of the reference's `Data_Import.LocInfo` (same attribute names; the
reference class cannot be constructed in this image: it needs openpyxl and calls
`pd.read_excel(sheetname=...)`, which pandas 2.x rejects).

Input files (`data_dir`, default parasitoids_amd/data): the reference's plain-text geometry files
`<site>fields.txt`, `<site>releasegrid.txt`, and CSV exports of its xlsx sheets made by
tests/golden/make_locinfo_fixtures.py (same cell contents, dates as ISO strings).

Only the Kalbar data set is wired up, like in the reference: every `location` branch of its
LocInfo other than 'kalbar' raises NotImplementedError (Data_Import.py:440,:491,:519,:561,:584,
:634), and `data/` holds field geometry and observation sheets for Kalbar only (Carnarvon has
wind and `carnarvonearlemergence.txt`, read by ParasitoidModel.emergence_data).
The reference's loader cannot run here, so no golden output exists; what the reference itself
holds for it are the assertions of its `test_LocInfo` / `test_model_emergence` /
`test_model_sampling` (tests/test_Bayes.py:39-230), which tests/test_locinfo_reference.py
replays against this class, next to direct reductions of the CSV files.
"""
import math
import os

import numpy as np
import pandas as pd

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_DATA_DIR = os.path.join(_HERE, 'data')

# the release grid is aligned with a nearby road (Data_Import.py:102-105)
GRID_ROTATION_DEG = -33.0


def latlong_tocoord(center, lat, long):
    '''(x east, y north) in metres of a lat/long point relative to `center`, equirectangular
    approximation on a sphere of radius 6378100 m (Data_Import.py:282-302).'''
    R = 6378100
    o_lat, o_long = math.radians(center[0]), math.radians(center[1])
    lat, long = math.radians(lat), math.radians(long)
    return (R * (long - o_long) * math.cos((o_lat + lat) / 2), R * (lat - o_lat))


def _data_lines(filename):
    '''lines with `#` comments removed; blank lines are kept (they separate polygons)'''
    with open(filename, 'r') as f:
        for line in f:
            c = line.find('#')
            yield (line if c < 0 else line[:c]).strip()


def read_field_polygons(filename, center):
    '''dict id -> matplotlib Path of the field boundary, vertices in metres from the release
    point (Data_Import.py:262-337, `LocInfo.get_fields`): an identifier line, then one
    `lat,long` vertex per line, fields separated by blank lines.'''
    from matplotlib.path import Path
    polys, verts, fid = {}, [], None

    def close(verts):
        codes = [Path.MOVETO] + [Path.LINETO] * (len(verts) - 1) + [Path.CLOSEPOLY]
        return Path(list(verts) + [(0., 0.)], codes)

    for line in _data_lines(filename):
        if line == '':
            if verts:
                polys[fid] = close(verts)
            verts, fid = [], None
        elif fid is None:
            fid = line
        else:
            lat, lon = line.split(',')[:2]
            verts.append(latlong_tocoord(center, float(lat), float(lon)))
    if verts:
        polys[fid] = close(verts)
    return polys


def field_cells(polys, domain_info):
    '''dict id -> int array [(row, col), ...] of the cells whose centre lies inside the field
    polygon (Data_Import.py:343-368; same point-in-polygon routine, matplotlib Path).'''
    R = int(domain_info[1])
    res = domain_info[0] / domain_info[1]
    colmesh, rowmesh = np.meshgrid(res * np.arange(-R, R + 1), res * np.arange(R, -R - 1, -1))
    centers = np.array([colmesh.flatten(), rowmesh.flatten()]).T
    out = {}
    for fid, path in polys.items():
        out[fid] = np.argwhere(path.contains_points(centers).reshape(2 * R + 1, 2 * R + 1))
    return out


def read_release_grid(filename):
    '''DataFrame xcoord, ycoord, area, samples, collection (Data_Import.py:372-415)'''
    rows = [[float(v) for v in line.split(',')] for line in _data_lines(filename) if line != '']
    data = np.array(rows)
    assert data.ndim == 2 and data.shape[1] == 5, 'incomplete line in {}'.format(filename)
    return pd.DataFrame(data, columns=['xcoord', 'ycoord', 'area', 'samples', 'collection'])


class LocInfo(object):
    '''field_polys, field_cells, field_sizes; grid_data, grid_cells; release_date,
    collection_datesPR, sent_DataFrames, sent_ids; releasefield_id, release_DataFrames,
    emerg_grids; grid_obs_DataFrame, grid_obs_datesPR, grid_obs, grid_samples;
    card_obs_DataFrames, card_obs_datesPR, step_size, card_obs; release_emerg,
    release_collection, sentinel_emerg -- as documented in Data_Import.py:14-51.'''

    def __init__(self, location, release_latlong, domain_info, data_dir=None):
        if location != 'kalbar':
            raise NotImplementedError(location)
        d = DEFAULT_DATA_DIR if data_dir is None else data_dir
        R = int(domain_info[1])
        res = domain_info[0] / domain_info[1]
        th = GRID_ROTATION_DEG / 180 * math.pi
        rot = np.array([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]])

        def rotate(frame):
            xy = frame[['xcoord', 'ycoord']].values
            out = np.array([rot @ p for p in xy]).reshape(-1, 2)
            frame['xcoord'], frame['ycoord'] = out[:, 0], out[:, 1]

        ##### sentinel fields (Data_Import.py:62-72)
        self.field_polys = read_field_polygons(os.path.join(d, location + 'fields.txt'), release_latlong)
        self.field_cells = field_cells(self.field_polys, domain_info)
        self.field_sizes = {k: int(max(v.shape)) for k, v in self.field_cells.items()}

        ##### release-field grid (Data_Import.py:74-113)
        self.grid_data = read_release_grid(os.path.join(d, location + 'releasegrid.txt'))
        rotate(self.grid_data)
        cells = np.array([-self.grid_data['ycoord'].values, self.grid_data['xcoord'].values])
        self.grid_cells = (np.around(cells / res) + R).T.astype(int)   # col 0 = row, col 1 = column

        ##### sentinel-field emergence (Data_Import.py:453-503)
        self.release_date = pd.Timestamp('2005-03-13')
        self.collection_datesPR = [pd.Timestamp('2005-03-31') - self.release_date]
        sen = pd.read_csv(os.path.join(d, 'kalbar_sentinels_raw.csv'), parse_dates=['date emerged'])
        sen = sen.rename(columns={'date emerged': 'date', 'Field ID (jpgs)': 'id'})
        sen = sen.drop(columns=['Field descrip', 'Field ID (paper)'])
        sen = sen.sort_values(['id', 'date'])
        counts = [c for c in sen.columns if c not in ('id', 'date')]
        sen['All_total'] = sen[counts].sum(axis=1)
        sen['E_total'] = sen[['Efemales', 'Emales']].sum(axis=1)
        sen['datePR'] = sen['date'] - self.release_date
        sen = sen.sort_values(['datePR', 'id']).reset_index(drop=True)
        self.sent_DataFrames = [sen]
        self.sent_ids = list(sen['id'].unique())

        ##### release-field emergence (Data_Import.py:525-563, :139-160)
        self.releasefield_id = 'A'
        rel = pd.read_csv(os.path.join(d, 'kalbar_releasefield_raw.csv'), parse_dates=['date emerged'])
        # north was on the left of the field grid: swap and flip, then move the release point
        # to the origin
        x_old = rel['xcoord'].copy()
        rel['xcoord'] = rel['ycoord'] - 200
        rel['ycoord'] = -x_old + 300
        counts = [c for c in rel.columns if c not in ('Field', 'xcoord', 'ycoord', 'date emerged')]
        rel['All_total'] = rel[counts].sum(axis=1)
        rel['E_total'] = rel[['Efemales', 'Emales']].sum(axis=1)
        rel['datePR'] = rel['date emerged'] - self.release_date
        rel = rel[(rel['xcoord'] != 0) & (rel['ycoord'] != 0)].reset_index()
        rel = rel.astype({'xcoord': float, 'ycoord': float})
        rotate(rel)
        rel['row'] = ((-rel['ycoord'] / res).round(0) + R).astype(int)
        rel['column'] = ((rel['xcoord'] / res).round(0) + R).astype(int)
        rel = rel.sort_values(['datePR', 'row', 'column']).reset_index(drop=True)
        self.release_DataFrames = [rel]
        self.emerg_grids = []
        for frame in self.release_DataFrames:
            first = frame['datePR'] == frame['datePR'].min()
            self.emerg_grids.append(list(zip(frame['row'][first].values, frame['column'][first].values)))

        ##### adult counts on the release-field grid (Data_Import.py:586-615, :162-191)
        obs = pd.read_csv(os.path.join(d, 'kalbar_adult_counts_field_A.csv'), parse_dates=['date'])
        obs = obs.rename(columns={'x coor': 'x', 'y coor': 'y', 'num leaves viewed': 'leaves',
                                  'num hayati': 'obs_count'})
        obs = obs[['date', 'collector', 'x', 'y', 'leaves', 'obs_count']].copy()
        obs['xcoord'] = obs['y'].astype(float) - 200
        obs['ycoord'] = -obs['x'].astype(float) + 300
        obs = obs.drop(columns=['x', 'y'])
        obs['datePR'] = obs['date'] - self.release_date
        obs = obs.sort_values(['datePR', 'xcoord', 'ycoord']).reset_index(drop=True)
        self.grid_obs_datesPR = [pd.Timedelta(t) for t in obs['datePR'].unique()]
        rotate(obs)
        self.grid_obs_DataFrame = obs
        ngrid = self.grid_cells.shape[0]
        self.grid_obs = np.zeros((ngrid, len(self.grid_obs_datesPR)))
        self.grid_samples = np.zeros((ngrid, len(self.grid_obs_datesPR)))
        gx, gy = self.grid_data['xcoord'].values, self.grid_data['ycoord'].values
        for nday, date in enumerate(self.grid_obs_datesPR):
            day = obs[obs['datePR'] == date]
            for n in range(ngrid):
                self.grid_samples[n, nday] = self.grid_data['samples'].iloc[n]
                hit = day[(day['xcoord'] == gx[n]) & (day['ycoord'] == gy[n])]
                if not hit.empty:
                    assert len(hit) == 1          # one count per grid point and date
                    self.grid_obs[n, nday] = hit['obs_count'].values[0]
        self.grid_samples = self.grid_samples / self.grid_samples.max()

        ##### adult counts along the cardinal directions (Data_Import.py:639-651, :193-216)
        self.step_size = [2, 2]
        self.card_obs_DataFrames, self.card_obs_datesPR, self.card_obs = [], [], []
        for name in ('kalbar_cardinal_15mar05.csv', 'kalbar_cardinal_21mar05.csv'):
            card = pd.read_csv(os.path.join(d, name), parse_dates=['date'])
            card = card.rename(columns={'num adults': 'obs_count'}).drop(columns=['num viewers'])
            card['datePR'] = card['date'] - self.release_date
            self.card_obs_datesPR.append(card['datePR'].iloc[0])
            card = card.sort_values(['direction', 'distance'])
            self.card_obs_DataFrames.append(card)
            sides = [card[card['direction'] == s]['obs_count'].values
                     for s in ('north', 'south', 'east', 'west')]
            arr = np.zeros((4, max(v.size for v in sides)))
            for i, v in enumerate(sides):
                arr[i, :v.size] = v
            self.card_obs.append(arr)

        ##### arrays shaped like popdensity_to_emergence's output (Data_Import.py:218-254)
        self.release_emerg, self.release_collection, self.sentinel_emerg = [], [], []
        for frame in self.release_DataFrames:
            dates = frame['datePR'].unique()
            first = frame['datePR'] == frame['datePR'].min()
            effort = []
            for x, y in frame.loc[first, ['xcoord', 'ycoord']].values:
                val = self.grid_data[(self.grid_data['xcoord'] == x) &
                                     (self.grid_data['ycoord'] == y)]['collection'].values
                assert val.shape == (1,)      # every collection point is a grid point, once
                effort.append(val[0])
            effort = np.array(effort)
            self.release_collection.append(effort / effort.max())
            E = np.zeros((int(first.sum()), len(dates)))
            for nd, date in enumerate(dates):
                E[:, nd] = frame[frame['datePR'] == date]['E_total'].values
            self.release_emerg.append(E)
        for frame in self.sent_DataFrames:
            dates = frame['datePR'].unique()
            E = np.zeros((len(self.sent_ids), len(dates)))
            for nd, date in enumerate(dates):
                E[:, nd] = frame[frame['datePR'] == date]['E_total'].values
            self.sentinel_emerg.append(E)
