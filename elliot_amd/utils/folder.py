"""Folder helpers (elliot/utils/folder.py:36-40 `build_model_folder`)."""
import os


def build_model_folder(path_output_rec_weight, model):
    os.makedirs(os.path.abspath(os.sep.join([path_output_rec_weight, model])), exist_ok=True)
