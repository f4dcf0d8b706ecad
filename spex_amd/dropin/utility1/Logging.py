"""Console + file logger with the reference's `Logging(filename).record(text)` surface
(LightGCN_SPEX/code/utility1/Logging.py): every record is echoed and appended as one CRLF-terminated line."""


class Logging:
    def __init__(self, filename):
        self.filename = filename

    def record(self, str_log):
        print(str_log)
        with open(self.filename, "a", newline="") as sink:
            print(str_log, file=sink, end="\r\n", flush=True)
