"""`python experiments.py test1_nobn_bilin_both train` -- the reference's launch line (README.md:77),
served by gan_heightmaps_amd.experiments on the MI355X backend."""
import sys

from gan_heightmaps_amd.experiments import main

if __name__ == '__main__':
    main(sys.argv)
