"""Stand-in for Engine in the ranks that launch.EngineFollowers starts (NVX_FOLLOWER_ENGINE=follower_standin:RecordingEngine):
writes what it is told, one JSON line per call, to $NVX_STANDIN_LOG.<rank>.  No device, no model."""
import json
import os


class RecordingEngine:
    def __init__(self, **kwargs):
        self.path = f"{os.environ['NVX_STANDIN_LOG']}.{os.environ['RANK']}"
        self._log({"call": "init", "kwargs": kwargs, "world": os.environ["WORLD_SIZE"], "master": os.environ["MASTER_ADDR"]})

    def _log(self, rec):
        with open(self.path, "a") as f:
            f.write(json.dumps(rec, sort_keys=True) + "\n")

    def add_sequence(self, sequence_id, prompt_token_ids, sampling_params):
        self._log({"call": "add", "id": sequence_id, "ids": prompt_token_ids, "temperature": sampling_params.temperature,
                   "max_new_tokens": sampling_params.max_new_tokens, "ignore_eos": sampling_params.ignore_eos})

    def abort_sequence(self, sequence_id):
        self._log({"call": "abort", "id": sequence_id})

    def step_overlapped(self):
        self._log({"call": "step"})
        return []

    def reset(self):
        self._log({"call": "reset"})

    def shutdown(self):
        self._log({"call": "shutdown"})
