-- helpers shared by several lens scripts: require "examples.lenses.shared.optics"
local M = {}
function M.stereographic_angle(r) return 2 * atan(r / 2) end
function M.tilt(deg)
   local a = deg * pi / 180
   return cos(a), sin(a)
end
return M
