"""Drop-in modules for the NGCF drivers (NGCF_SPEX/code): `ngcf_parser`, `utility.load_data`, `utility.batch_test`,
`utility.helper`, `utility.metrics`, `utility.Logging` — the import names NGCF_SPEX/code/main_rec.py:2,14-16 uses.
`python -m spex_amd.dropin <driver>` puts this directory first on sys.path when the driver imports `ngcf_parser`
(the LightGCN modules one level up answer to `lg_parser`); the model class the reference defines inside its driver
(`Model_Wrapper`, main_rec.py:36-113) is `spex_amd.ngcf.Model_Wrapper`."""
import os

PATH = os.path.dirname(os.path.abspath(__file__))
