"""Logging in the reference's line format "[L::module::func] message"
(xcltk/utils/xlog.py:7-54), so logs of the two implementations diff cleanly."""
import logging

_LEVEL_CHAR = {logging.DEBUG: "D", logging.INFO: "I", logging.WARNING: "W",
               logging.ERROR: "E", logging.CRITICAL: "C"}


class XFormatter(logging.Formatter):
    def __init__(self, datefmt=None):
        super().__init__(fmt=None, datefmt=datefmt)

    def format(self, record):
        parts = [_LEVEL_CHAR.get(record.levelno, "U")]
        if record.module:
            parts.append(record.module)
        if record.funcName:
            parts.append(record.funcName)
        if self.datefmt:
            parts.append(self.formatTime(record, self.datefmt))
        text = "[%s] %s" % ("::".join(parts), record.getMessage())
        if record.exc_info and not record.exc_text:
            record.exc_text = self.formatException(record.exc_info)
        for extra in (record.exc_text, self.formatStack(record.stack_info) if record.stack_info else None):
            if extra:
                text = text + ("" if text.endswith("\n") else "\n") + extra
        return text


def init_logging(log_file=None, stream=None, fh_level=logging.DEBUG, fh_datefmt="%Y-%m-%d %H:%M:%S",
                 ch_level=logging.INFO, ch_datefmt=None):
    if log_file is None and stream is None:
        raise ValueError("at least one of 'log_file' and 'stream' should not be None.")
    handlers = []
    if log_file:
        fh = logging.FileHandler(log_file, mode="w")
        fh.setLevel(fh_level)
        fh.setFormatter(XFormatter(datefmt=fh_datefmt))
        handlers.append(fh)
    if stream:
        ch = logging.StreamHandler(stream=stream)
        ch.setLevel(ch_level)
        ch.setFormatter(XFormatter(datefmt=ch_datefmt))
        handlers.append(ch)
    logging.basicConfig(level=logging.DEBUG, handlers=handlers)
