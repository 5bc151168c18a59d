-- shim: `require 'dpnn'` (models.lua:4): required by the reference, but no dpnn symbol is used on the path (SURVEY.md 8c)
return {}
