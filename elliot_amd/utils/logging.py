"""Logger lookup used by the plugin surface.

Inside a real Elliot process the loggers are created by `logging_project.prepare_logger(key, ...)`
(elliot/run.py:66) and looked up with `get_logger_model` (elliot/utils/logging.py:77-84); we defer to it when
Elliot is importable.  Stand-alone (tests, bench, our own runner) plain `logging` loggers are used.
"""
import logging as pylog

try:  # pragma: no cover - only inside an Elliot installation
    from elliot.utils import logging as _elog
except Exception:  # Elliot (or one of its deps) not importable: stand-alone mode
    _elog = None


def get_logger_model(name, log_level=pylog.DEBUG):
    if _elog is not None:
        # Elliot registers plugin loggers under the YAML key (run.py:66), i.e. "external.<Class>" for us
        for cand in (name, f"external.{name}"):
            try:
                return _elog.get_logger_model(cand, log_level)
            except Exception:
                pass
    logger = pylog.getLogger(f"elliot_amd.{name}")
    logger.setLevel(log_level)
    return logger


def get_logger(name, log_level=pylog.DEBUG):
    if _elog is not None:
        try:
            return _elog.get_logger(name, log_level)
        except Exception:
            pass
    logger = pylog.getLogger(f"elliot_amd.{name}")
    logger.setLevel(log_level)
    return logger
