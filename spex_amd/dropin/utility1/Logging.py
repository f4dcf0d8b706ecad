"""Console + file logger with the reference's `Logging(filename).record(text)` surface
(LightGCN_SPEX/code/utility1/Logging.py): every record is echoed and appended to the file as one CRLF-terminated
line."""
import sys


class Logging:
    def __init__(self, filename):
        self.filename = filename

    def record(self, str_log):
        line = str(str_log)
        sys.stdout.write(line + "\n")
        with open(self.filename, "a", newline="") as sink:
            sink.write(line + "\r\n")
