"""Dimensions and split definitions of the DB2/DB3 Ninapro setup.

Values follow /root/reference/code/constants.py:1-97 (names kept so code written
against the reference reads the same); they are recomputed here from the same
seed-0 draws of numpy's legacy generator, not copied.
"""
import numpy as np

_rs = np.random.RandomState(0)          # code/constants.py:3  (np.random.seed(0))

MAX_PEOPLE_D2 = 40                      # code/constants.py:5,9
MAX_PEOPLE_D3 = 6                       # code/constants.py:6,10   subjects [2,3,4,5,8,9] of DB3
MAX_PEOPLE = MAX_PEOPLE_D2 + MAX_PEOPLE_D3

d2_idxs = _rs.permutation(MAX_PEOPLE_D2)             # code/constants.py:18
d3_idxs = _rs.permutation(MAX_PEOPLE_D3)             # code/constants.py:19
PEOPLE_IDXS = np.concatenate((d2_idxs, d3_idxs + len(d2_idxs)))   # code/constants.py:24
TRAIN_PEOPLE_IDXS = PEOPLE_IDXS
TEST_PEOPLE_IDXS = PEOPLE_IDXS

TASKS_A = np.arange(1, 18, dtype=np.uint8)           # exercise B grasps, code/constants.py:37
TASKS_B = np.arange(18, 41, dtype=np.uint8)          # exercise C grasps, code/constants.py:38
_rs.shuffle(TASKS_A)
_rs.shuffle(TASKS_B)
TASKS = np.concatenate((TASKS_A, TASKS_B))
TEST_TASKS = TASKS[:]
TRAIN_TASKS = TASKS[:]
TASK_DIST = np.array([17, 23])
MAX_TASKS = int(TASK_DIST.sum()) + 1                 # 41, rest included
MAX_TASKS_TRAIN = MAX_TASKS

REPS = [1, 3, 4, 6, 2, 5]                            # code/constants.py:50
MAX_REPS = len(REPS)
REPS_TRAIN = REPS[:-2]
REPS_TEST = REPS[-2:]

Hz = 2000
DOWNSAMPLE = 100
FACTOR = Hz // DOWNSAMPLE
RMS_WINDOW = 11
WINDOW_EDGE = (RMS_WINDOW - 1) // 2
TOTAL_WINDOW_SIZE = Hz
FINAL_WINDOW_SIZE = TOTAL_WINDOW_SIZE // FACTOR      # 100 samples per repetition

VOTE = True
PREDICTION_WINDOW = 250                              # ms
PREDICTION_WINDOW_SIZE = PREDICTION_WINDOW * DOWNSAMPLE // 1000      # 25 samples
AMT_PREDICTION_WINDOWS = FINAL_WINDOW_SIZE // PREDICTION_WINDOW_SIZE  # 4 runs per repetition
assert FINAL_WINDOW_SIZE % AMT_PREDICTION_WINDOWS == 0

WINDOW_MS = 1
WINDOW_STRIDE = 1
WINDOW_OUTPUT_DIM = FINAL_WINDOW_SIZE
AMT_WINDOWS = FINAL_WINDOW_SIZE // WINDOW_MS

GLOVE_DIM = 22 - 2
EMG_DIM = 12
