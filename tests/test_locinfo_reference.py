"""The reference's own checks of `Data_Import.LocInfo` and of the model -> observation
translation (its tests/test_Bayes.py:39-230: test_LocInfo, test_model_emergence,
test_model_sampling), replayed against parasitoids_amd.Data_Import.LocInfo.  They are the
reference-held pins for this loader (its own cannot run here: openpyxl / read_excel(sheetname=)).
Fixture values are the reference's: Kalbar, centre (-27.945752, 152.58474), default
domain_info (8000.0, 320) (conftest.py:37-44)."""
import warnings

import numpy as np
import pandas as pd
import pytest
from matplotlib.path import Path

DOMAIN_INFO = (8000.0, 320)


@pytest.fixture(scope='module')
def locinfo():
    from parasitoids_amd.Data_Import import LocInfo
    return LocInfo('kalbar', (-27.945752, 152.58474), DOMAIN_INFO)


def test_LocInfo(locinfo):
    '''tests/test_Bayes.py:39-133, assertion for assertion'''
    ### Field boundary information ###
    assert type(locinfo.field_polys) is dict
    assert type(locinfo.field_polys['A']) is Path
    assert len(locinfo.field_polys) == 7                      # Fields: A, B, C, D, E, F, G
    assert type(locinfo.field_cells) is dict
    assert isinstance(locinfo.field_cells['A'], np.ndarray)
    assert len(locinfo.field_cells) == 7
    assert type(locinfo.field_sizes) is dict
    assert type(locinfo.field_sizes['A']) is int
    assert len(locinfo.field_sizes) == 7

    ### Release field grid info ###
    assert isinstance(locinfo.grid_data, pd.DataFrame)
    for key in ['xcoord', 'ycoord', 'samples', 'collection']:
        assert key in locinfo.grid_data.keys()
    assert isinstance(locinfo.grid_cells, np.ndarray)
    assert locinfo.grid_cells.shape[1] == 2
    assert locinfo.grid_data['xcoord'].size == locinfo.grid_cells.shape[0]

    ### Sentinel field emergence data ###
    assert isinstance(locinfo.release_date, pd.Timestamp)
    assert isinstance(locinfo.collection_datesPR, list)
    assert isinstance(locinfo.collection_datesPR[0], pd.Timedelta)
    assert locinfo.collection_datesPR[0] > pd.Timedelta('0 days')
    assert isinstance(locinfo.sent_DataFrames[0], pd.DataFrame)
    for key in ['id', 'datePR', 'E_total', 'All_total']:
        assert key in locinfo.sent_DataFrames[0].keys()
    assert np.all(locinfo.sent_DataFrames[0]['E_total'].values <=
                  locinfo.sent_DataFrames[0]['All_total'].values)
    for key in locinfo.sent_ids:
        assert key in locinfo.field_cells.keys()
    minTimedelta = locinfo.collection_datesPR[0]
    for Td in locinfo.sent_DataFrames[0]['datePR']:
        assert Td >= minTimedelta

    ### Release field emergence data ###
    assert isinstance(locinfo.releasefield_id, str)
    for key in ['row', 'column', 'xcoord', 'ycoord', 'datePR', 'E_total', 'All_total']:
        assert key in locinfo.release_DataFrames[0].keys()
    for coord in locinfo.release_DataFrames[0][['xcoord', 'ycoord']].values:
        assert coord in locinfo.grid_data[['xcoord', 'ycoord']].values
    assert np.all(locinfo.release_DataFrames[0]['E_total'].values <=
                  locinfo.release_DataFrames[0]['All_total'].values)
    for Td in locinfo.release_DataFrames[0]['datePR']:
        assert Td >= minTimedelta
    grid_cells_list = locinfo.grid_cells.tolist()
    for cell in locinfo.release_DataFrames[0][['row', 'column']].values.tolist():
        assert cell in grid_cells_list
        assert tuple(cell) in locinfo.emerg_grids[0]

    ### Grid observation data ###
    assert isinstance(locinfo.grid_obs_DataFrame, pd.DataFrame)
    assert isinstance(locinfo.grid_obs_datesPR, list)
    assert isinstance(locinfo.grid_obs_datesPR[0], pd.Timedelta)
    assert isinstance(locinfo.grid_obs, np.ndarray)
    assert isinstance(locinfo.grid_samples, np.ndarray)
    assert np.all(locinfo.grid_obs.shape == locinfo.grid_samples.shape)
    assert locinfo.grid_samples.max() == 1
    assert locinfo.grid_obs.max() > 0

    ### Cardinal direction data ###
    assert isinstance(locinfo.card_obs_DataFrames, list)
    assert isinstance(locinfo.card_obs_DataFrames[0], pd.DataFrame)
    assert isinstance(locinfo.card_obs_datesPR, list)
    assert isinstance(locinfo.card_obs_datesPR[0], pd.Timedelta)
    assert isinstance(locinfo.step_size, list)
    assert isinstance(locinfo.card_obs, list)
    assert isinstance(locinfo.card_obs[0], np.ndarray)
    assert len(locinfo.card_obs_DataFrames) == len(locinfo.card_obs_datesPR) \
        == len(locinfo.step_size) == len(locinfo.card_obs)
    for c_obs in locinfo.card_obs:
        assert c_obs.shape[0] == 4


def test_locinfo_pymc_structures_are_consistent(locinfo):
    '''the parts of tests/test_Bayes.py:137-195 that need no model solution: the PyMC-friendly
    arrays against the DataFrames they are built from'''
    for ii in range(len(locinfo.release_DataFrames)):
        frame = locinfo.release_DataFrames[ii]
        n_obs = len(frame['datePR'].unique())
        assert locinfo.release_emerg[ii].shape == (len(locinfo.emerg_grids[ii]), n_obs)
        assert locinfo.release_collection[ii].size == len(locinfo.emerg_grids[ii])
        sen = locinfo.sent_DataFrames[ii]
        assert locinfo.sentinel_emerg[ii].shape == (len(locinfo.sent_ids), len(sen['datePR'].unique()))
        # the grid points match from data frame to emerg_grids on every emergence day
        for n, cell in enumerate(locinfo.emerg_grids[ii]):
            for day in frame['datePR'].unique():
                assert tuple(frame[frame['datePR'] == day][['row', 'column']].values[n, :]) == cell
        # same for sentinel fields
        for n, field in enumerate(locinfo.sent_ids):
            for day in sen['datePR'].unique():
                assert sen[sen['datePR'] == day]['id'].values[n] == field
        # release_collection should be relative numbers
        assert locinfo.release_collection[ii].max() == 1
        assert locinfo.release_collection[ii].min() >= 0


@pytest.mark.gpu
def test_model_emergence_and_sampling(locinfo):
    '''tests/test_Bayes.py:137-230 (test_model_emergence, test_model_sampling) with the model
    solution coming from the device-resident population model instead of a saved output file'''
    from parasitoids_amd import Bayes_funcs as Bayes
    from parasitoids_amd import ParasitoidModel as PM
    from parasitoids_amd.pop_model import PopModel
    from helpers import HP, DP, DLP, MU_R, NPER
    wd, days = PM.get_wind_data('data/kalbar', 30, '00:00')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        modelsol = PopModel(wd, days, domain_info=DOMAIN_INFO, r_number=130000)
        modelsol.evaluate(HP, DP, DLP, MU_R, NPER)
    release_emerg, sentinel_emerg = Bayes.popdensity_to_emergence(modelsol, locinfo)
    assert isinstance(release_emerg, list)
    for ii in range(len(release_emerg)):
        n_grid_pts, n_obs = release_emerg[ii].shape
        assert n_grid_pts == len(locinfo.emerg_grids[ii])
        assert n_obs == len(locinfo.release_DataFrames[ii]['datePR'].unique())
        assert n_grid_pts == locinfo.release_emerg[ii].shape[0]
        assert n_obs == locinfo.release_emerg[ii].shape[1]
        assert n_grid_pts == locinfo.release_collection[ii].size
        n_fields, n_obs = sentinel_emerg[ii].shape
        assert n_fields == len(locinfo.sent_ids)
        assert n_obs == len(locinfo.sent_DataFrames[ii]['datePR'].unique())
        assert (n_fields, n_obs) == locinfo.sentinel_emerg[ii].shape
    grid_counts = Bayes.popdensity_grid(modelsol, locinfo)
    card_counts = Bayes.popdensity_card(modelsol, locinfo, DOMAIN_INFO)
    assert np.all(grid_counts.shape == locinfo.grid_obs.shape == locinfo.grid_samples.shape)
    assert grid_counts.max() > 0
    assert grid_counts.min() >= 0
    for nobs, obs in enumerate(locinfo.card_obs):
        assert np.all(obs.shape == card_counts[nobs].shape)
        assert card_counts[nobs].max() > 0
        assert card_counts[nobs].min() >= 0
    modelsol.close()
