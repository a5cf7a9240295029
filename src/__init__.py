"""Drop-in module paths of the reference (`python -m src.mnist --train|--sample`,
`python -m src.shakespeare ...`, `src.utils`): thin aliases of
tinydiffusionmodels_amd, which holds the MI355X-native implementation."""
